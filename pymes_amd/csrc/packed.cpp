// Packed binary integral file (see packed.h).  Host code only.
#include "packed.h"

#include <sys/types.h>

#include <cerrno>
#include <cstring>
#include <stdexcept>

namespace pymes {

int64_t packed_block_doubles(int pattern, int no, int nv) {
    int64_t s = 1;
    for (int i = 0; i < 4; ++i) s *= (pattern >> (3 - i) & 1) ? nv : no;
    return s;
}

uint64_t packed_payload_doubles(int kind, int n_orb, int n_occ, int naux) {
    if (kind == kPackedFactors) return static_cast<uint64_t>(naux) * n_orb * n_orb;
    uint64_t tot = 0;
    for (int pat = 0; pat < 16; ++pat) tot += static_cast<uint64_t>(packed_block_doubles(pat, n_occ, n_orb - n_occ));
    return tot;       // = n_orb^4
}

static void put_header(unsigned char* raw, const PackedHeader& hd) {
    std::memset(raw, 0, kPackedHeaderBytes);
    std::memcpy(raw, hd.magic, 8);
    const int32_t ints[8] = {hd.kind, hd.n_orb, hd.n_elec, hd.n_occ, hd.naux, 0, 0, 0};
    std::memcpy(raw + 8, ints, sizeof ints);
    std::memcpy(raw + 40, &hd.e_core, 8);
    std::memcpy(raw + 48, &hd.payload_doubles, 8);
}
static void get_header(const unsigned char* raw, PackedHeader& hd) {
    std::memcpy(hd.magic, raw, 8);
    int32_t ints[8];
    std::memcpy(ints, raw + 8, sizeof ints);
    hd.kind = ints[0]; hd.n_orb = ints[1]; hd.n_elec = ints[2]; hd.n_occ = ints[3]; hd.naux = ints[4];
    std::memcpy(&hd.e_core, raw + 40, 8);
    std::memcpy(&hd.payload_doubles, raw + 48, 8);
}

PackedReader::PackedReader(const std::string& path) : path_(path) {
    fp_.reset(std::fopen(path.c_str(), "rb"));
    if (!fp_) throw std::runtime_error("cannot open " + path + ": " + std::strerror(errno));
    unsigned char raw[kPackedHeaderBytes];
    if (std::fread(raw, 1, sizeof raw, fp_.get()) != sizeof raw) throw std::runtime_error(path + ": truncated header");
    get_header(raw, head);
    if (std::memcmp(head.magic, "PYMESPK1", 8) != 0) throw std::runtime_error(path + ": not a PYMESPK1 packed integral file");
    if (head.kind != kPackedBlocks && head.kind != kPackedFactors) throw std::runtime_error(path + ": unknown payload kind");
    if (head.n_orb < 2 || head.n_occ < 1 || head.n_occ >= head.n_orb || head.n_elec != 2 * head.n_occ)
        throw std::runtime_error(path + ": inconsistent orbital counts in the header");
    if (head.kind == kPackedFactors && head.naux < 1) throw std::runtime_error(path + ": factors need naux >= 1");
    if (head.payload_doubles != packed_payload_doubles(head.kind, head.n_orb, head.n_occ, head.naux))
        throw std::runtime_error(path + ": payload size does not match the header");
    const size_t n = static_cast<size_t>(head.n_orb);
    eps.resize(n);
    h.resize(n * n);
    if (std::fread(eps.data(), 8, n, fp_.get()) != n || std::fread(h.data(), 8, n * n, fp_.get()) != n * n)
        throw std::runtime_error(path + ": truncated one-body part");
    // the file must hold exactly the payload
    const off_t here = ftello(fp_.get());
    fseeko(fp_.get(), 0, SEEK_END);
    const off_t end = ftello(fp_.get());
    fseeko(fp_.get(), here, SEEK_SET);
    if (here < 0 || end < here || static_cast<uint64_t>(end - here) != 8 * head.payload_doubles)
        throw std::runtime_error(path + ": file length does not match the payload size");
}
void PackedReader::read(double* dst, uint64_t doubles) {
    if (std::fread(dst, 8, doubles, fp_.get()) != doubles) throw std::runtime_error(path_ + ": truncated payload");
}

PackedWriter::PackedWriter(const std::string& path, int kind, int n_orb, int n_elec, int naux, double e_core,
                           const double* eps, const double* h)
    : path_(path) {
    if (n_elec % 2 || n_elec < 2 || n_elec / 2 >= n_orb) throw std::runtime_error("packed write: need an even NELEC with 1 <= NELEC/2 < NORB");
    PackedHeader hd{};
    std::memcpy(hd.magic, "PYMESPK1", 8);
    hd.kind = kind; hd.n_orb = n_orb; hd.n_elec = n_elec; hd.n_occ = n_elec / 2; hd.naux = kind == kPackedFactors ? naux : 0;
    hd.e_core = e_core;
    hd.payload_doubles = expected_ = packed_payload_doubles(kind, n_orb, hd.n_occ, naux);
    fp_.reset(std::fopen(path.c_str(), "wb"));
    if (!fp_) throw std::runtime_error("cannot open " + path + " for writing: " + std::strerror(errno));
    unsigned char raw[kPackedHeaderBytes];
    put_header(raw, hd);
    const size_t n = static_cast<size_t>(n_orb);
    if (std::fwrite(raw, 1, sizeof raw, fp_.get()) != sizeof raw || std::fwrite(eps, 8, n, fp_.get()) != n ||
        std::fwrite(h, 8, n * n, fp_.get()) != n * n)
        throw std::runtime_error(path + ": write failed");
}
void PackedWriter::write(const double* src, uint64_t doubles) {
    if (std::fwrite(src, 8, doubles, fp_.get()) != doubles) throw std::runtime_error(path_ + ": write failed");
    written_ += doubles;
}
void PackedWriter::close() {
    if (written_ != expected_) throw std::runtime_error(path_ + ": incomplete payload");
    if (std::fclose(fp_.release()) != 0) throw std::runtime_error(path_ + ": close failed");
}

}  // namespace pymes
