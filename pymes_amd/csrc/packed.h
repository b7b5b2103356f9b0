// Packed binary integral file ("PYMESPK1"): the on-disk counterpart of the 16 partition.py blocks (or of density-fitting
// factors), read straight into the device blocks.  The reference has only the FCIDUMP text format, parsed line by
// line in Python (pymes/util/fcidump.py:124-161) and the hdf5 branch of its TCDUMP reader (pymes/util/tcdump.py:44-48,
// 88-92); a (50 occ, 200 virt) problem is ~1e9 text lines there.  Host code only (no device calls).
//
// Layout (little-endian):
//   header, 64 bytes: char magic[8] = "PYMESPK1"; int32 kind (1 = blocks, 2 = factors); int32 n_orb; int32 n_elec;
//                     int32 n_occ; int32 naux (kind 2, else 0); int32 reserved[3]; double e_core; uint64 payload_doubles
//   double eps[n_orb]; double h[n_orb][n_orb];
//   kind 1: the 16 blocks of V[p,q,r,s] = <pq|rs> in pattern order 0..15 (bit 3-pos set = index `pos` virtual: 0 = klij,
//           3 = ijab, 12 = abij, 15 = abcd), each C-contiguous in its own index order;
//   kind 2: B[naux][n_orb][n_orb] with V[p,q,r,s] = sum_Q B[Q,p,r] B[Q,q,s].
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

namespace pymes {

struct PackedHeader {
    char magic[8];
    int32_t kind, n_orb, n_elec, n_occ, naux, reserved[3];
    double e_core;
    uint64_t payload_doubles;
};
static_assert(sizeof(PackedHeader) == 56 || sizeof(PackedHeader) == 64, "unexpected header packing");

// closes on destruction — also when a constructor throws after the fopen (members are destroyed, the destructor is not run)
struct FileCloser {
    void operator()(FILE* f) const { if (f) std::fclose(f); }
};
using FileHandle = std::unique_ptr<FILE, FileCloser>;

constexpr int kPackedBlocks = 1, kPackedFactors = 2;
constexpr size_t kPackedHeaderBytes = 64;

int64_t packed_block_doubles(int pattern, int no, int nv);
uint64_t packed_payload_doubles(int kind, int n_orb, int n_occ, int naux);

// An open packed file positioned at its payload.  Throws std::runtime_error (bad magic, truncated file, size mismatch).
class PackedReader {
  public:
    explicit PackedReader(const std::string& path);
    PackedHeader head{};
    std::vector<double> eps, h;
    void read(double* dst, uint64_t doubles);      // next `doubles` values of the payload

  private:
    FileHandle fp_;
    std::string path_;
};

class PackedWriter {
  public:
    PackedWriter(const std::string& path, int kind, int n_orb, int n_elec, int naux, double e_core, const double* eps,
                 const double* h);
    void write(const double* src, uint64_t doubles);
    void close();                                   // checks that the whole payload was written

  private:
    FileHandle fp_;
    std::string path_;
    uint64_t expected_ = 0, written_ = 0;
};

}  // namespace pymes
