// Native FCIDUMP text parser with the semantics of pymes/util/fcidump.py:59-163 (host code, no device calls).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pymes {

struct FcidumpFile {
    int n_elec = 0, n_orb = 0;
    double e_core = 0.0;
    std::vector<double> eps;        // [n]
    std::vector<double> h;          // [n,n]
    // two-electron lines in file order, 0-based, already renamed (i j k l) -> (p r q s):  val, p, q, r, s
    std::vector<double> val;
    std::vector<int32_t> pqrs;      // 4 per line
};

// Throws std::runtime_error with the reference's failure modes: unterminated header, a body line that does not
// have exactly five fields (a blank line included), fields that are not numbers, orbital indices beyond NORB.
void parse_fcidump(const std::string& path, FcidumpFile& out, bool header_only = false);

// V[n,n,n,n] (zero-initialised by the caller) filled line by line in file order exactly like fcidump.py:140-149
void fill_V_host(const FcidumpFile& f, bool is_tc, double* V);

}  // namespace pymes
