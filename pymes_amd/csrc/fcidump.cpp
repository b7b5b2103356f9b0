// Native FCIDUMP text parser (pymes/util/fcidump.py:59-163).  Host code only.
#include "fcidump.h"

#include <algorithm>
#include <charconv>
#include <functional>
#include <thread>
#include <system_error>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace pymes {
namespace {

std::string lower(std::string s) {
    for (auto& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    return s;
}
std::string strip(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace(static_cast<unsigned char>(s[a]))) ++a;
    while (b > a && std::isspace(static_cast<unsigned char>(s[b - 1]))) --b;
    return s.substr(a, b - a);
}
bool all_digits(const std::string& s) {
    if (s.empty()) return false;
    for (char c : s)
        if (!std::isdigit(static_cast<unsigned char>(c))) return false;
    return true;
}

// fcidump.py:100-116: the header is every line up to the first one containing '/' or "END" (case-insensitive), with the
// lines stripped and concatenated; NORB / NELEC are found by substring match on the comma-separated fields.
void parse_header(const std::string& head, FcidumpFile& out) {
    size_t pos = 0;
    while (pos <= head.size()) {
        size_t comma = head.find(',', pos);
        if (comma == std::string::npos) comma = head.size();
        const std::string field = head.substr(pos, comma - pos), low = lower(field);
        for (int which = 0; which < 2; ++which) {
            if (low.find(which == 0 ? "norb" : "nelec") == std::string::npos) continue;
            size_t p2 = 0;
            while (p2 <= field.size()) {
                size_t eq = field.find('=', p2);
                if (eq == std::string::npos) eq = field.size();
                const std::string word = strip(field.substr(p2, eq - p2));
                if (all_digits(word)) (which == 0 ? out.n_orb : out.n_elec) = std::atoi(word.c_str());
                p2 = eq + 1;
            }
        }
        pos = comma + 1;
    }
}

}  // namespace

void parse_fcidump(const std::string& path, FcidumpFile& out, bool header_only) {
    FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("cannot open " + path + ": " + std::strerror(errno));
    std::string buf;
    {
        char chunk[1 << 16];
        size_t got;
        // the header sits in the first lines: a header-only query (to size the context) does not read the body
        while ((got = std::fread(chunk, 1, sizeof chunk, fp)) > 0) {
            buf.append(chunk, got);
            if (header_only && buf.size() >= (size_t(1) << 20)) break;
        }
        std::fclose(fp);
    }
    // ---- header ---------------------------------------------------------------------------------------------
    size_t pos = 0;
    std::string head;
    bool closed = false;
    while (pos < buf.size()) {
        size_t nl = buf.find('\n', pos);
        if (nl == std::string::npos) nl = buf.size();
        head += strip(buf.substr(pos, nl - pos));
        pos = std::min(nl + 1, buf.size());
        if (head.find('/') != std::string::npos || lower(head).find("end") != std::string::npos) {
            closed = true;
            break;
        }
    }
    if (!closed) throw std::runtime_error("FCIDUMP header is not terminated by '/' or '&END'");
    out = FcidumpFile();
    parse_header(head, out);
    if (header_only) return;
    const int n = out.n_orb;
    out.eps.assign(static_cast<size_t>(n), 0.0);
    out.h.assign(static_cast<size_t>(n) * n, 0.0);
    // ---- body: "value i j k l", exactly five fields per line (fcidump.py:124-161) ----------------------------------
    // The body is cut into pieces at line boundaries and the pieces are parsed by worker threads (the reference walks
    // the file in one Python loop); records are joined in file order, so "a later line overwrites an earlier one" holds
    // exactly as in the sequential reader, and the first malformed line of the FILE is the one reported.
    const char* body = buf.c_str() + pos;
    const char* end = buf.c_str() + buf.size();
    struct OneBody { long p, q, r, s; double v; };
    struct Piece {
        const char* b; const char* e;
        std::vector<double> val;
        std::vector<int32_t> pqrs;
        std::vector<OneBody> small;      // core energy / orbital energies / h_pq records, in order
        long lines = 0, bad_line = 0;    // bad_line: 1-based within the piece
        std::string error;
    };
    const size_t nbytes = static_cast<size_t>(end - body);
    unsigned nthreads = 1;
    if (nbytes > (size_t(4) << 20)) {
        nthreads = std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
        if (const char* e = std::getenv("PYMES_PARSE_THREADS")) nthreads = std::max(1, std::atoi(e));
    }
    std::vector<Piece> pieces(nthreads);
    {
        const char* cur = body;
        for (unsigned t = 0; t < nthreads; ++t) {
            const char* stop = (t + 1 == nthreads) ? end : body + nbytes * (t + 1) / nthreads;
            if (stop < cur) stop = cur;
            if (t + 1 < nthreads && stop < end) {          // move to the end of the line the cut fell into
                const char* nl = static_cast<const char*>(std::memchr(stop, '\n', static_cast<size_t>(end - stop)));
                stop = nl ? nl + 1 : end;
            }
            pieces[t].b = cur;
            pieces[t].e = stop;
            cur = stop;
        }
    }
    auto parse_piece = [n](Piece& pc) {
        const char* p = pc.b;
        pc.val.reserve(static_cast<size_t>(pc.e - pc.b) / 40 + 16);
        pc.pqrs.reserve(4 * (static_cast<size_t>(pc.e - pc.b) / 40 + 16));
        auto fail = [&](const std::string& what) { pc.bad_line = pc.lines; pc.error = what; };
        while (p < pc.e) {
            const char* eol = static_cast<const char*>(std::memchr(p, '\n', static_cast<size_t>(pc.e - p)));
            if (!eol) eol = pc.e;
            ++pc.lines;
            const char* tok[6];
            size_t len[6];
            int nt = 0;
            const char* q = p;
            while (q < eol && nt < 6) {
                while (q < eol && std::isspace(static_cast<unsigned char>(*q))) ++q;
                if (q >= eol) break;
                tok[nt] = q;
                while (q < eol && !std::isspace(static_cast<unsigned char>(*q))) ++q;
                len[nt] = static_cast<size_t>(q - tok[nt]);
                ++nt;
            }
            if (nt != 5) {
                fail(": expected 5 fields, got " + std::to_string(nt == 6 ? 6 : nt) + (nt == 0 ? " (blank line)" : ""));
                return;
            }
            // value: std::from_chars (correctly rounded, no locale, no copy); anything it does not take whole goes to strtod
            double v = 0.0;
            {
                const char* f0 = tok[0] + (tok[0][0] == '+' ? 1 : 0);
                auto res = std::from_chars(f0, tok[0] + len[0], v);
                if (res.ec != std::errc() || res.ptr != tok[0] + len[0]) {
                    char tmp[64];
                    if (len[0] >= sizeof tmp) { fail(": bad value"); return; }
                    std::memcpy(tmp, tok[0], len[0]);
                    tmp[len[0]] = 0;
                    char* stop = nullptr;
                    v = std::strtod(tmp, &stop);
                    if (stop == tmp || *stop) { fail(": bad value"); return; }
                }
            }
            long id[4];
            for (int i = 0; i < 4; ++i) {
                const char* f0 = tok[i + 1] + (tok[i + 1][0] == '+' ? 1 : 0);
                auto res = std::from_chars(f0, tok[i + 1] + len[i + 1], id[i], 10);
                if (res.ec != std::errc() || res.ptr != tok[i + 1] + len[i + 1]) { fail(": bad index"); return; }
            }
            p = eol + 1;
            if (std::fabs(v) < 1e-19) continue;                                        // :138
            const long pp = id[0] - 1, rr = id[1] - 1, qq = id[2] - 1, ss = id[3] - 1;   // i j k l -> p r q s (:130)
            if (pp >= n || qq >= n || rr >= n || ss >= n) { fail(": orbital index beyond NORB"); return; }
            if (pp >= 0 && qq >= 0 && rr >= 0 && ss >= 0) {
                pc.val.push_back(v);
                pc.pqrs.push_back(static_cast<int32_t>(pp));
                pc.pqrs.push_back(static_cast<int32_t>(qq));
                pc.pqrs.push_back(static_cast<int32_t>(rr));
                pc.pqrs.push_back(static_cast<int32_t>(ss));
            } else {
                pc.small.push_back({pp, qq, rr, ss, v});
            }
        }
    };
    if (nthreads == 1) {
        parse_piece(pieces[0]);
    } else {
        std::vector<std::thread> workers;
        for (unsigned t = 0; t < nthreads; ++t) workers.emplace_back(parse_piece, std::ref(pieces[t]));
        for (auto& w : workers) w.join();
    }
    long before = 0;
    size_t total = 0;
    for (auto& pc : pieces) {
        if (!pc.error.empty()) {
            const std::string where = std::to_string(before + pc.bad_line);
            if (pc.error.find("beyond NORB") != std::string::npos) throw std::runtime_error("FCIDUMP line " + where + pc.error);
            throw std::runtime_error("malformed FCIDUMP line " + where + pc.error);
        }
        before += pc.lines;
        total += pc.val.size();
    }
    out.val.reserve(total);
    out.pqrs.reserve(4 * total);
    for (auto& pc : pieces) {
        out.val.insert(out.val.end(), pc.val.begin(), pc.val.end());
        out.pqrs.insert(out.pqrs.end(), pc.pqrs.begin(), pc.pqrs.end());
        for (const auto& r : pc.small) {
            if (r.p < 0 && r.q < 0 && r.r < 0 && r.s < 0) {
                out.e_core = r.v;                                                  // :151-152
            } else if (r.p >= 0 && r.q < 0 && r.r < 0 && r.s < 0) {
                out.eps[r.p] = r.v;                                                // :154-155
            } else if (r.p >= 0 && r.r >= 0 && r.q < 0 && r.s < 0) {
                out.h[static_cast<size_t>(r.p) * n + r.r] = r.v;                   // :157-160
                out.h[static_cast<size_t>(r.r) * n + r.p] = r.v;
            }
        }
    }
}

void fill_V_host(const FcidumpFile& f, bool is_tc, double* V) {
    const int64_t n = f.n_orb;
    auto at = [&](int64_t a, int64_t b, int64_t c, int64_t d) -> double& { return V[((a * n + b) * n + c) * n + d]; };
    for (size_t t = 0; t < f.val.size(); ++t) {
        const int64_t p = f.pqrs[4 * t], q = f.pqrs[4 * t + 1], r = f.pqrs[4 * t + 2], s = f.pqrs[4 * t + 3];
        const double x = f.val[t];
        if (is_tc) {                                                               // :148-149
            at(q, p, s, r) = x;
            at(p, q, r, s) = x;
        } else {                                                                   // :143-146
            at(p, q, r, s) = x;
            at(r, q, p, s) = x;
            at(r, s, p, q) = x;
            at(p, s, r, q) = x;
        }
    }
}

}  // namespace pymes
