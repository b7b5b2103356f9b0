// Native FCIDUMP text parser (pymes/util/fcidump.py:59-163).  Host code only.
#include "fcidump.h"

#include <algorithm>
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
        while ((got = std::fread(chunk, 1, sizeof chunk, fp)) > 0) buf.append(chunk, got);
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
    const char* p = buf.c_str() + pos;
    const char* end = buf.c_str() + buf.size();
    long lineno = 0;
    while (p < end) {
        const char* eol = static_cast<const char*>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        if (!eol) eol = end;
        ++lineno;
        // tokenise
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
            throw std::runtime_error("malformed FCIDUMP line " + std::to_string(lineno) + ": expected 5 fields, got " +
                                     std::to_string(nt == 6 ? 6 : nt) + (nt == 0 ? " (blank line)" : ""));
        }
        char tmp[64];
        auto field = [&](int i) {
            if (len[i] >= sizeof tmp) throw std::runtime_error("malformed FCIDUMP line " + std::to_string(lineno));
            std::memcpy(tmp, tok[i], len[i]);
            tmp[len[i]] = 0;
            return tmp;
        };
        char* stop = nullptr;
        const double v = std::strtod(field(0), &stop);
        if (stop == tmp || *stop) throw std::runtime_error("malformed FCIDUMP line " + std::to_string(lineno) + ": bad value");
        long id[4];
        for (int i = 0; i < 4; ++i) {
            id[i] = std::strtol(field(i + 1), &stop, 10);
            if (stop == tmp || *stop) throw std::runtime_error("malformed FCIDUMP line " + std::to_string(lineno) + ": bad index");
        }
        p = eol + 1;
        if (std::fabs(v) < 1e-19) continue;                                        // :138
        const long pp = id[0] - 1, rr = id[1] - 1, qq = id[2] - 1, ss = id[3] - 1;   // i j k l -> p r q s (:130)
        if (pp >= n || qq >= n || rr >= n || ss >= n)
            throw std::runtime_error("FCIDUMP line " + std::to_string(lineno) + ": orbital index beyond NORB");
        if (pp >= 0 && qq >= 0 && rr >= 0 && ss >= 0) {
            out.val.push_back(v);
            out.pqrs.push_back(static_cast<int32_t>(pp));
            out.pqrs.push_back(static_cast<int32_t>(qq));
            out.pqrs.push_back(static_cast<int32_t>(rr));
            out.pqrs.push_back(static_cast<int32_t>(ss));
        } else if (pp < 0 && qq < 0 && rr < 0 && ss < 0) {
            out.e_core = v;                                                        // :151-152
        } else if (pp >= 0 && qq < 0 && rr < 0 && ss < 0) {
            out.eps[pp] = v;                                                       // :154-155
        } else if (pp >= 0 && rr >= 0 && qq < 0 && ss < 0) {
            out.h[static_cast<size_t>(pp) * n + rr] = v;                           // :157-160
            out.h[static_cast<size_t>(rr) * n + pp] = v;
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
