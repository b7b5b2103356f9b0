// Host engine: arena, tensor views, TTGT contraction planner, integral blocks.
#include "engine.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>

namespace pymes {

// ---------------------------------------------------------------------------------
// views
// ---------------------------------------------------------------------------------
TView make_view(double* p, int rank, const int64_t* dims, const int64_t* strides) {
    if (rank < 0 || rank > 6) throw Error("tensor rank must be 0..6");
    TView t;
    t.p = p;
    t.rank = rank;
    int64_t s = 1;
    for (int i = rank - 1; i >= 0; --i) {
        t.dim[i] = dims[i];
        t.st[i] = strides ? strides[i] : s;
        s *= dims[i];
    }
    return t;
}
TView make_view(double* p, std::initializer_list<int64_t> dims) {
    int64_t d[6];
    int r = 0;
    for (auto x : dims) {
        if (r >= 6) throw Error("tensor rank must be 0..6");
        d[r++] = x;
    }
    return make_view(p, r, d, nullptr);
}
TView make_view(const double* p, std::initializer_list<int64_t> dims) {
    return make_view(const_cast<double*>(p), dims);
}
TView slice(const TView& t, int axis, int64_t lo, int64_t hi) {
    if (axis < 0 || axis >= t.rank || lo < 0 || hi > t.dim[axis] || lo > hi) throw Error("bad slice");
    TView r = t;
    r.p = t.p + lo * t.st[axis];
    r.dim[axis] = hi - lo;
    return r;
}

// ---------------------------------------------------------------------------------
// arena
// ---------------------------------------------------------------------------------
void Arena::init(size_t bytes) {
    release();
    base_ = static_cast<char*>(dev::dmalloc(bytes));
    cap_ = bytes;
    top_ = high_ = 0;
}
void Arena::release() {
    if (base_) dev::dfree(base_);
    base_ = nullptr;
    cap_ = top_ = 0;
    deferred_ = false;
}
void Arena::reset(size_t m) {
    if (m >= top_) return;
    // (only while the arena is at most half full: a hole under live allocations comes back when an enclosing scope ends, and
    // the engine's own high-water mark is a quarter of the capacity it asks for — above that the release is immediate, as ever)
    if (dev::phase_pending() && top_ <= cap_ / 2) {
        want_ = deferred_ ? std::min(want_, m) : m;
        deferred_ = true;
        gen_ = dev::phase_generation();
        return;
    }
    top_ = m;
    deferred_ = false;
}
double* Arena::alloc(int64_t doubles) {
    size_t bytes = (static_cast<size_t>(std::max<int64_t>(doubles, 1)) * sizeof(double) + 255) & ~size_t(255);
    if (deferred_) {
        if (!dev::phase_pending() || dev::phase_generation() != gen_) {
            top_ = want_;                          // the tasks recorded when the release was made have been launched
        } else if (top_ + bytes > cap_) {          // out of room: launch the phase, take the release
            dev::phase_sync();
            top_ = want_;
        }
        deferred_ = false;                         // (else: above the hole; an enclosing scope's release takes it back)
    }
    if (top_ + bytes > cap_)
        throw Error("workspace exhausted: need " + std::to_string((top_ + bytes) >> 20) + " MiB, have " +
                    std::to_string(cap_ >> 20) + " MiB (pass a larger workspace_bytes to pymes_ctx_create)");
    double* p = reinterpret_cast<double*>(base_ + top_);
    top_ += bytes;
    high_ = std::max(high_, top_);
    return p;
}

// ---------------------------------------------------------------------------------
// block names
// ---------------------------------------------------------------------------------
int pattern_of_name(const char* name) {
    if (!name || std::strlen(name) != 4) throw Error("block name must have 4 letters");
    int pat = 0;
    for (int i = 0; i < 4; ++i) {
        const char c = name[i];
        if (c >= 'a' && c <= 'd') pat |= 1 << (3 - i);
        else if (!(c >= 'i' && c <= 'l')) throw Error(std::string("bad block name: ") + name);
    }
    return pat;
}
std::string canonical_name(int pattern) {
    // spellings of pymes/integral/partition.py:4-39
    static const char* names[16] = {"klij", "ijka", "ijak", "ijab", "iajk", "iajb", "iabj", "iabc",
                                    "aijk", "aijb", "aibj", "aibc", "abij", "abic", "abci", "abcd"};
    return names[pattern & 15];
}

// ---------------------------------------------------------------------------------
// engine basics
// ---------------------------------------------------------------------------------
Engine::Engine(int device_, int no_, int nv_, size_t workspace_bytes) : device(device_), no(no_), nv(nv_), n(no_ + nv_) {
    if (no < 1 || nv < 1) throw Error("need at least one occupied and one virtual orbital");
    dev::set_device(device);
    own_stream_ = dev::stream_create();
    stream = own_stream_;
    if (workspace_bytes == 0) {
        // ~24 amplitude-sized temporaries + the largest dressing temporary (o v^3) + slack
        const double o = no, v = nv;
        double bytes = 8.0 * (24.0 * o * o * v * v + 2.0 * o * v * v * v + 4.0 * (o + v) * (o + v) * o * o) +
                       (64 << 20);
        workspace_bytes = static_cast<size_t>(bytes);
    }
    arena.init(workspace_bytes);
    splitk_doubles_ = 32 << 20;   // 256 MiB of split-K partials (2048 tile-splits of 128x128)
    splitk_ws_ = static_cast<double*>(dev::dmalloc(sizeof(double) * splitk_doubles_));
    eps_o = static_cast<double*>(dev::dmalloc(sizeof(double) * no));
    eps_v = static_cast<double*>(dev::dmalloc(sizeof(double) * nv));
}

Engine::~Engine() {
    try {
        dev::set_device(device);
        if (capturing_) dev::graph_abort(stream);
        dev::stream_sync(stream);
        for (auto g : graphs_) dev::graph_destroy(g);
        for (void* p : user_allocs_) dev::dfree(p);
        for (auto& kv : scratch_free_) dev::dfree(kv.second);
        for (auto& kv : scratch_live_) dev::dfree(kv.first);
        for (auto& p : lay_) dev::dfree(p);
        dev::dfree(xs_oo_);
        dev::dfree(xs_vv_);
        for (auto& p : V_) dev::dfree(p);
        for (auto& p : Vd_) dev::dfree(p);
        for (auto& kv : static_) dev::dfree(kv.second);
        dev::dfree(splitk_ws_);
        dev::dfree(lpack_.Vp);
        dev::dfree(lpack_.Vm);
        dev::dfree(eps_o);
        dev::dfree(eps_v);
        arena.release();
        dev::stream_destroy(own_stream_);
    } catch (...) {
    }
}


double* Engine::scratch_get(int64_t doubles) {
    doubles = std::max<int64_t>(doubles, 1);
    auto it = scratch_free_.find(doubles);
    double* p;
    if (it != scratch_free_.end()) {
        p = it->second;
        scratch_free_.erase(it);
    } else {
        p = static_cast<double*>(dev::try_dmalloc(sizeof(double) * static_cast<size_t>(doubles)));
        if (!p) {
            scratch_trim();
            try {
                p = static_cast<double*>(dev::dmalloc(sizeof(double) * static_cast<size_t>(doubles)));
            } catch (const std::exception&) {
                throw Error("out of device memory (" + std::to_string((doubles * 8) >> 20) + " MiB of scratch)");
            }
        }
    }
    scratch_live_[p] = doubles;
    return p;
}
void Engine::scratch_put(double* p) {
    auto it = scratch_live_.find(p);
    if (it == scratch_live_.end()) return;
    scratch_free_.emplace(it->second, p);
    scratch_live_.erase(it);
}
void Engine::scratch_trim() {
    if (scratch_free_.empty()) return;
    dev::stream_sync(stream);
    for (auto& kv : scratch_free_) dev::dfree(kv.second);
    scratch_free_.clear();
}
int64_t Engine::scratch_free_bytes() const {
    int64_t b = 0;
    for (auto& kv : scratch_free_) b += kv.first * 8;
    return b;
}

void* Engine::user_malloc(size_t bytes) {
    void* p = dev::dmalloc(bytes);
    user_allocs_.insert(p);
    return p;
}
void Engine::user_free(void* p) {
    if (!p) return;
    auto it = user_allocs_.find(p);
    if (it == user_allocs_.end()) throw Error("pymes_free: not a live allocation of this context");
    if (capturing_) throw Error("pymes_free while a launch graph is being recorded");
    dev::stream_sync(stream);
    dev::dfree(p);
    user_allocs_.erase(it);
}

void Engine::graph_begin() {
    if (capturing_) throw Error("graph_begin: already recording");
    if (!dev::graphs_supported()) throw Error("launch graphs are not supported by this backend");
    dev::graph_begin(stream);
    capturing_ = true;
    capture_generation_ = dress_generation_;
}
dev::graph_t Engine::graph_end() {
    if (!capturing_) throw Error("graph_end: not recording");
    capturing_ = false;
    dev::graph_t g = nullptr;
    try {
        g = dev::graph_end(stream);
    } catch (...) {
        dev::graph_abort(stream);
        throw;
    }
    graphs_.insert(g);
    if (dress_generation_ != capture_generation_) graphs_dressing_.insert(g);
    return g;
}
void Engine::graph_abort() {
    if (!capturing_) return;
    capturing_ = false;
    dev::graph_abort(stream);
}
void Engine::graph_launch(dev::graph_t g) {
    if (!graphs_.count(g)) throw Error("graph_launch: unknown graph");
    if (graphs_dressing_.count(g)) ++dress_generation_;
    dev::graph_launch(g, stream);
}
void Engine::graph_destroy(dev::graph_t g) {
    auto it = graphs_.find(g);
    if (it == graphs_.end()) return;
    dev::stream_sync(stream);
    dev::graph_destroy(g);
    graphs_.erase(it);
    graphs_dressing_.erase(g);
    if (graphs_.empty() && release_wanted_) release_residual_buffers();
}

void Engine::exchange_asymmetry_V(double out[2]) {
    out[0] = out[1] = 0.0;
    for (int pat = 0; pat < 16; ++pat) {
        if (!V_[pat]) continue;
        // partner block under (p,q,r,s) -> (q,p,s,r): swap the type bits of the two bras and of the two kets
        const int partner = ((pat & 8) >> 1) | ((pat & 4) << 1) | ((pat & 2) >> 1) | ((pat & 1) << 1);
        if (!V_[partner]) {
            out[0] = 1.0 / 0.0;
            continue;
        }
        if (partner < pat) continue;          // each pair once
        int64_t d[4];
        for (int i = 0; i < 4; ++i) d[i] = (pat >> (3 - i) & 1) ? nv : no;
        double r[2];
        dev::exchange_asymmetry(V_[pat], V_[partner], d, r, stream);
        out[0] = std::max(out[0], r[0]);
        out[1] = std::max(out[1], r[1]);
    }
}

void Engine::invalidate_static() {
    for (auto& kv : static_) dev::dfree(kv.second);
    static_.clear();
    lpack_.valid = false;
}

int64_t Engine::block_size(int pattern) const {
    int64_t s = 1;
    for (int i = 0; i < 4; ++i) s *= (pattern >> (3 - i) & 1) ? nv : no;
    return s;
}
TView Engine::block_view(double* p, int pattern) const {
    int64_t d[4];
    for (int i = 0; i < 4; ++i) d[i] = (pattern >> (3 - i) & 1) ? nv : no;
    return make_view(p, 4, d, nullptr);
}
bool Engine::has_block(int pattern, bool dressed) const {
    return (dressed ? Vd_[pattern & 15] : V_[pattern & 15]) != nullptr;
}
TView Engine::block(int pattern, bool dressed) {
    double* p = dressed ? Vd_[pattern & 15] : V_[pattern & 15];
    if (!p)
        throw Error(std::string(dressed ? "dressed" : "undressed") + " integral block '" + canonical_name(pattern) +
                    "' has not been set");
    return block_view(p, pattern);
}
double* Engine::ensure_block(int pattern) {
    pattern &= 15;
    invalidate_static();
    if (!V_[pattern]) V_[pattern] = static_cast<double*>(dev::dmalloc(sizeof(double) * block_size(pattern)));
    return V_[pattern];
}
double* Engine::ensure_dressed(int pattern) {
    if (!Vd_[pattern]) Vd_[pattern] = static_cast<double*>(dev::dmalloc(sizeof(double) * block_size(pattern)));
    return Vd_[pattern];
}

void Engine::set_orbital_energies(const double* eo_host, const double* ev_host) {
    dev::memcpy_h2d(eps_o, eo_host, sizeof(double) * no, stream);
    dev::memcpy_h2d(eps_v, ev_host, sizeof(double) * nv, stream);
    eps_set = true;
}

void Engine::need_eps(const char* who) const {
    if (!eps_set) throw Error(std::string(who) + ": the orbital energies have not been set (pymes_set_orbital_energies)");
}

// ---------------------------------------------------------------------------------
// permute / axpby
// ---------------------------------------------------------------------------------
static void check_labels(const char* s, int rank, const char* what) {
    if (static_cast<int>(std::strlen(s)) != rank)
        throw Error(std::string("label string '") + s + "' does not match rank of " + what);
    for (int i = 0; i < rank; ++i)
        for (int j = i + 1; j < rank; ++j)
            if (s[i] == s[j]) throw Error(std::string("repeated label in '") + s + "'");
}

static bool is_contiguous(const TView& t) {
    int64_t expect = 1;
    for (int i = t.rank - 1; i >= 0; --i) {
        if (t.dim[i] != 1 && t.st[i] != expect) return false;
        expect *= t.dim[i];
    }
    return true;
}

void Engine::permute(double alpha, const TView& in, const char* si, double beta, const TView& out,
                     const char* so) {
    check_labels(si, in.rank, "input");
    check_labels(so, out.rank, "output");
    if (in.rank != out.rank) throw Error("permute: rank mismatch");
    dev::Permute p;
    p.rank = out.rank;
    p.alpha = alpha;
    p.beta = beta;
    p.in = in.p;
    p.out = out.p;
    for (int i = 0; i < out.rank; ++i) {
        const char* f = std::strchr(si, so[i]);
        if (!f) throw Error(std::string("permute: label '") + so[i] + "' missing from input '" + si + "'");
        const int j = static_cast<int>(f - si);
        if (in.dim[j] != out.dim[i]) throw Error(std::string("permute: extent mismatch for label '") + so[i] + "'");
        p.dim[i] = out.dim[i];
        p.s_in[i] = in.st[j];
        p.s_out[i] = out.st[i];
    }
    for (int i = out.rank; i < 6; ++i) {
        p.dim[i] = 1;
        p.s_in[i] = p.s_out[i] = 0;
    }
    dev::permute(p, stream);
    stats.permute_calls++;
    stats.permute_bytes += 8.0 * static_cast<double>(out.size()) * (beta != 0.0 ? 3.0 : 2.0);
}

void Engine::axpby(double alpha, const TView& in, double beta, const TView& out) {
    static const char* lab = "abcdef";
    if (in.rank != out.rank) throw Error("axpby: rank mismatch");
    std::string s(lab, lab + in.rank);
    permute(alpha, in, s.c_str(), beta, out, s.c_str());
}

void Engine::zero(const TView& t) {
    // contiguous fast path, else scale-by-zero copy of itself
    bool contig = true;
    int64_t s = 1;
    for (int i = t.rank - 1; i >= 0; --i) {
        if (t.dim[i] != 1 && t.st[i] != s) contig = false;
        s *= t.dim[i];
    }
    if (contig) dev::memset_zero(t.p, sizeof(double) * t.size(), stream);
    else axpby(0.0, t, 0.0, t);
}

// ---------------------------------------------------------------------------------
// contraction planner
// ---------------------------------------------------------------------------------
namespace {

struct Label {
    char c;
    int64_t n;
    int pa, pb, pc;   // positions in A, B, C or -1
    char kind;        // 'M', 'N', 'K', 'Z' (batch)
};

// merge the dims of X at positions `pos` (outer -> inner) into one (size, stride)
bool merge_group(const TView& X, const std::vector<int>& pos, int64_t& size, int64_t& stride) {
    size = 1;
    stride = 0;
    bool have = false;
    for (int t = static_cast<int>(pos.size()) - 1; t >= 0; --t) {
        const int64_t nd = X.dim[pos[t]], s = X.st[pos[t]];
        if (nd == 1) continue;
        if (!have) {
            size = nd;
            stride = s;
            have = true;
        } else {
            if (s != size * stride) return false;
            size *= nd;
        }
    }
    return true;
}

struct Operand {
    const TView* v;
    const char* s;
};

}  // namespace

void Engine::contract(double alpha, const TView& A, const char* sa, const TView& B, const char* sb, double beta,
                      const TView& C, const char* sc, const char* batch, const TView* Cin) {
    if (Cin) {
        if (Cin->rank != C.rank) throw Error("contract: Cin must have the shape of C");
        for (int i = 0; i < C.rank; ++i)
            if (Cin->dim[i] != C.dim[i] || (C.dim[i] != 1 && Cin->st[i] != C.st[i]))
                throw Error("contract: Cin must have the shape and strides of C");
        if (beta == 0.0) Cin = nullptr;
    }
    check_labels(sa, A.rank, "A");
    check_labels(sb, B.rank, "B");
    check_labels(sc, C.rank, "C");
    const std::string spec = std::string(sa) + "," + sb + "->" + sc;
    // ---- classify labels ------------------------------------------------------------
    std::vector<Label> labs;
    auto find = [&](char c) -> Label* {
        for (auto& l : labs)
            if (l.c == c) return &l;
        return nullptr;
    };
    auto add = [&](const TView& X, const char* s, int which) {
        for (int i = 0; i < X.rank; ++i) {
            Label* l = find(s[i]);
            if (!l) {
                labs.push_back({s[i], X.dim[i], -1, -1, -1, '?'});
                l = &labs.back();
            }
            if (l->n != X.dim[i]) throw Error("contract " + spec + ": extent mismatch for label '" + s[i] + "'");
            (which == 0 ? l->pa : which == 1 ? l->pb : l->pc) = i;
        }
    };
    add(A, sa, 0);
    add(B, sb, 1);
    add(C, sc, 2);
    for (auto& l : labs) {
        const bool a = l.pa >= 0, b = l.pb >= 0, c = l.pc >= 0;
        const bool forced = std::strchr(batch, l.c) != nullptr;
        if (a && b && c) l.kind = 'Z';
        else if (a && c) l.kind = forced ? 'Z' : 'M';
        else if (b && c) l.kind = forced ? 'Z' : 'N';
        else if (a && b) l.kind = 'K';
        else throw Error("contract " + spec + ": label '" + l.c + "' appears in only one tensor");
        if (forced && !c) throw Error("contract " + spec + ": batch label '" + l.c + "' must be an output index");
    }
    // label lists ordered by position in a given tensor
    auto ordered = [&](char kind, int which) {
        std::vector<const Label*> r;
        for (auto& l : labs)
            if (l.kind == kind) r.push_back(&l);
        std::sort(r.begin(), r.end(), [&](const Label* x, const Label* y) {
            auto pos = [&](const Label* l) { return which == 0 ? l->pa : which == 1 ? l->pb : l->pc; };
            return pos(x) < pos(y);
        });
        return r;
    };
    auto positions = [&](const std::vector<const Label*>& ls, int which) {
        std::vector<int> r;
        for (auto* l : ls) r.push_back(which == 0 ? l->pa : which == 1 ? l->pb : l->pc);
        return r;
    };
    const std::vector<const Label*> M_cand[2] = {ordered('M', 2), ordered('M', 0)};
    const std::vector<const Label*> N_cand[2] = {ordered('N', 2), ordered('N', 1)};
    const std::vector<const Label*> K_cand[2] = {ordered('K', 0), ordered('K', 1)};
    const std::vector<const Label*> Z = ordered('Z', 2);

    struct Choice {
        int im, in_, ik;
        bool copyA, copyB, copyC;
        double cost;
    } best{0, 0, 0, true, true, true, 1e300};
    auto direct_ab = [&](const TView& X, int which, const std::vector<const Label*>& g1,
                         const std::vector<const Label*>& g2) {
        int64_t n1, s1, n2, s2;
        if (!merge_group(X, positions(g1, which), n1, s1)) return false;
        if (!merge_group(X, positions(g2, which), n2, s2)) return false;
        return s1 == 1 || s2 == 1 || n1 == 1 || n2 == 1;
    };
    // a transposed copy of a whole undressed integral block is made once and kept (make_copy below): it costs next to
    // nothing per call, so when one operand has to be copied the static one is the one to copy — not the amplitudes
    auto kept = [&](const TView& X) {
        for (int pat = 0; pat < 16; ++pat)
            if (V_[pat] && X.p == V_[pat] && X.size() == block_size(pat) && is_contiguous(X)) return true;
        return false;
    };
    const double bytesA = 8.0 * A.size() * (kept(A) ? 0.01 : 1.0), bytesB = 8.0 * B.size() * (kept(B) ? 0.01 : 1.0),
                 bytesC = 8.0 * C.size();
    for (int im = 0; im < 2; ++im)
        for (int in_ = 0; in_ < 2; ++in_)
            for (int ik = 0; ik < 2; ++ik) {
                Choice c{im, in_, ik, false, false, false, 0.0};
                c.copyA = !direct_ab(A, 0, M_cand[im], K_cand[ik]);
                c.copyB = !direct_ab(B, 1, K_cand[ik], N_cand[in_]);
                c.copyC = !direct_ab(C, 2, M_cand[im], N_cand[in_]);
                c.cost = (c.copyA ? 2 * bytesA : 0) + (c.copyB ? 2 * bytesB : 0) +
                         (c.copyC ? (beta != 0.0 ? 3 : 2) * bytesC : 0);
                if (c.cost < best.cost) best = c;
            }
    const auto& Ms = M_cand[best.im];
    const auto& Ns = N_cand[best.in_];
    const auto& Ks = K_cand[best.ik];

    if (Cin && best.copyC) {      // no fused path through a transposed temporary: do the copy explicitly
        copy(*Cin, C);
        Cin = nullptr;
    }
    ArenaScope scope(arena);
    // ---- materialise copies where needed ----------------------------------------------
    // canonical copy layout: [batch labels present (C order)][group1][group2], contiguous
    auto make_copy = [&](const TView& X, const char* sx, int which, const std::vector<const Label*>& g1,
                         const std::vector<const Label*>& g2, bool fill) {
        std::string lay;
        for (auto* l : Z) {
            const int pos = which == 0 ? l->pa : which == 1 ? l->pb : l->pc;
            if (pos >= 0) lay.push_back(l->c);
        }
        for (auto* l : g1) lay.push_back(l->c);
        for (auto* l : g2) lay.push_back(l->c);
        int64_t d[6];
        for (size_t i = 0; i < lay.size(); ++i) d[i] = find(lay[i])->n;
        // A transposed copy of a whole UNDRESSED integral block is the same in every iteration: keep it
        // (dressed Fock / singles terms contract o v^3-sized blocks with T1 over non-adjacent indices).
        if (fill) {
            for (int pat = 0; pat < 16; ++pat) {
                if (X.p != V_[pat] || !V_[pat] || X.size() != block_size(pat) || !is_contiguous(X)) continue;
                std::string key = "perm:" + std::to_string(pat) + ":";
                for (char ch : lay) key.push_back(static_cast<char>('0' + (std::strchr(sx, ch) - sx)));
                auto it = static_.find(key);
                if (it == static_.end()) {
                    const size_t bytes = sizeof(double) * static_cast<size_t>(X.size());
                    if (dev::mem_free_bytes() < 2 * bytes + (size_t(8) << 30)) break;     // not worth the memory
                    double* p = static_cast<double*>(dev::dmalloc(bytes));
                    it = static_.emplace(key, p).first;
                    permute(1.0, X, sx, 0.0, make_view(p, static_cast<int>(lay.size()), d, nullptr), lay.c_str());
                }
                return std::make_pair(make_view(it->second, static_cast<int>(lay.size()), d, nullptr), lay);
            }
        }
        TView t = make_view(arena.alloc(X.size()), static_cast<int>(lay.size()), d, nullptr);
        if (fill) permute(1.0, X, sx, 0.0, t, lay.c_str());
        return std::make_pair(t, lay);
    };
    TView Av = A, Bv = B, Cv = C;
    std::string la = sa, lb = sb, lc = sc;
    if (best.copyA) std::tie(Av, la) = make_copy(A, sa, 0, Ms, Ks, true);
    if (best.copyB) std::tie(Bv, lb) = make_copy(B, sb, 1, Ks, Ns, true);
    if (best.copyC) std::tie(Cv, lc) = make_copy(C, sc, 2, Ms, Ns, false);

    auto pos_in = [&](const std::string& lay, const std::vector<const Label*>& g) {
        std::vector<int> r;
        for (auto* l : g) r.push_back(static_cast<int>(lay.find(l->c)));
        return r;
    };
    int64_t Msz, Nsz, Ksz, a_sm, a_sk, b_sk, b_sn, c_sm, c_sn, tmp;
    if (!merge_group(Av, pos_in(la, Ms), Msz, a_sm) || !merge_group(Av, pos_in(la, Ks), Ksz, a_sk) ||
        !merge_group(Bv, pos_in(lb, Ks), tmp, b_sk) || !merge_group(Bv, pos_in(lb, Ns), Nsz, b_sn) ||
        !merge_group(Cv, pos_in(lc, Ms), tmp, c_sm) || !merge_group(Cv, pos_in(lc, Ns), tmp, c_sn))
        throw Error("contract " + spec + ": internal planner error (group not mergeable)");

    // ---- batch dims (C order), merged where all three stride sets allow ------------------
    struct BD {
        int64_t n, sa, sb, sc;
    };
    std::vector<BD> bd;
    for (auto* l : Z) {
        if (l->n == 1) continue;
        auto st = [&](const TView& X, const std::string& lay) -> int64_t {
            const size_t p = lay.find(l->c);
            return p == std::string::npos ? 0 : X.st[p];
        };
        BD d{l->n, st(Av, la), st(Bv, lb), st(Cv, lc)};
        if (!bd.empty()) {
            BD& o = bd.back();
            if (o.sa == d.n * d.sa && o.sb == d.n * d.sb && o.sc == d.n * d.sc) {
                o.n *= d.n;
                o.sa = d.sa;
                o.sb = d.sb;
                o.sc = d.sc;
                continue;
            }
        }
        bd.push_back(d);
    }
    while (bd.size() < 2) bd.insert(bd.begin(), BD{1, 0, 0, 0});
    const size_t nouter = bd.size() - 2;
    int64_t outer_total = 1;
    for (size_t i = 0; i < nouter; ++i) outer_total *= bd[i].n;

    dev::Gemm g;
    g.alpha = alpha;
    g.beta = best.copyC ? 0.0 : beta;
    g.K = Ksz;
    g.nb1 = bd[nouter].n;
    g.nb2 = bd[nouter + 1].n;
    g.splitk_ws = splitk_ws_;
    g.splitk_ws_doubles = splitk_doubles_;
    // orientation: the kernel writes C rows with unit stride along its N
    const bool transposed = !(c_sn == 1 || Nsz == 1) && (c_sm == 1 || Msz == 1);
    if (!(c_sn == 1 || Nsz == 1) && !transposed)
        throw Error("contract " + spec + ": internal planner error (C has no unit stride)");
    for (int64_t it = 0; it < outer_total; ++it) {
        int64_t rem = it, oa = 0, ob = 0, oc = 0;
        for (int i = static_cast<int>(nouter) - 1; i >= 0; --i) {
            const int64_t c = rem % bd[i].n;
            rem /= bd[i].n;
            oa += c * bd[i].sa;
            ob += c * bd[i].sb;
            oc += c * bd[i].sc;
        }
        if (!transposed) {
            g.M = Msz; g.N = Nsz;
            g.A = Av.p + oa; g.a_sm = a_sm; g.a_sk = a_sk;
            g.B = Bv.p + ob; g.b_sk = b_sk; g.b_sn = b_sn;
            g.C = Cv.p + oc; g.ldc = c_sm;
            g.a_b1 = bd[nouter].sa; g.a_b2 = bd[nouter + 1].sa;
            g.b_b1 = bd[nouter].sb; g.b_b2 = bd[nouter + 1].sb;
        } else {   // C^T = B^T A^T
            g.M = Nsz; g.N = Msz;
            g.A = Bv.p + ob; g.a_sm = b_sn; g.a_sk = b_sk;
            g.B = Av.p + oa; g.b_sk = a_sk; g.b_sn = a_sm;
            g.C = Cv.p + oc; g.ldc = c_sn;
            g.a_b1 = bd[nouter].sb; g.a_b2 = bd[nouter + 1].sb;
            g.b_b1 = bd[nouter].sa; g.b_b2 = bd[nouter + 1].sa;
        }
        g.c_b1 = bd[nouter].sc; g.c_b2 = bd[nouter + 1].sc;
        g.Cin = Cin ? Cin->p + oc : nullptr;
        dev::gemm(g, stream);
        stats.gemm_calls++;
        stats.gemm_flops += 2.0 * double(Msz) * double(Nsz) * double(Ksz) * double(g.nb1) * double(g.nb2);
    }
    if (best.copyC) permute(1.0, Cv, lc.c_str(), beta, C, sc);
    // a product that went through temporaries of this call's arena scope must not wait in an open group (dev::gemm_group_*):
    // the scope ends here and the next call may reuse the memory
    if (best.copyA || best.copyB || best.copyC) dev::gemm_group_sync();
}

// ---------------------------------------------------------------------------------
// integral blocks
// ---------------------------------------------------------------------------------
void Engine::set_V_full(const double* V, bool on_device, const int64_t strides[4]) {
    invalidate_static();
    const int64_t nn = n;
    double* full = nullptr;
    const double* src = V;
    int64_t st[4] = {nn * nn * nn, nn * nn, nn, 1};
    if (!on_device) {
        full = static_cast<double*>(dev::dmalloc(sizeof(double) * nn * nn * nn * nn));
        dev::memcpy_h2d(full, V, sizeof(double) * nn * nn * nn * nn, stream);
        src = full;
    } else if (strides) {
        for (int i = 0; i < 4; ++i) st[i] = strides[i];
    }
    try {
        for (int pat = 0; pat < 16; ++pat) {
            if (!V_[pat]) V_[pat] = static_cast<double*>(dev::dmalloc(sizeof(double) * block_size(pat)));
            TView dst = block_view(V_[pat], pat);
            TView sv;
            sv.rank = 4;
            int64_t off = 0;
            for (int i = 0; i < 4; ++i) {
                const bool virt = pat >> (3 - i) & 1;
                sv.dim[i] = virt ? nv : no;
                sv.st[i] = st[i];
                if (virt) off += no * st[i];
            }
            sv.p = const_cast<double*>(src) + off;
            copy(sv, dst);
        }
        dev::stream_sync(stream);
    } catch (...) {
        dev::dfree(full);
        throw;
    }
    dev::dfree(full);
}

void Engine::set_V_block(const char* name, const double* data, bool on_device, const int64_t strides[4]) {
    const int pat = pattern_of_name(name);
    invalidate_static();
    if (!V_[pat]) V_[pat] = static_cast<double*>(dev::dmalloc(sizeof(double) * block_size(pat)));
    if (!on_device) {
        dev::memcpy_h2d(V_[pat], data, sizeof(double) * block_size(pat), stream);
        return;
    }
    TView dst = block_view(V_[pat], pat);
    TView sv = dst;
    sv.p = const_cast<double*>(data);
    if (strides)
        for (int i = 0; i < 4; ++i) sv.st[i] = strides[i];
    copy(sv, dst);
}

void Engine::set_V_from_factors(const double* B_host, int naux) {
    // V[p,q,r,s] = (pr|qs) = sum_Q B[Q,p,r] B[Q,q,s]   (density-fitted / synthetic input)
    invalidate_static();
    if (naux < 1) throw Error("naux must be positive");
    const int64_t nn = n;
    double* Bd = static_cast<double*>(dev::dmalloc(sizeof(double) * naux * nn * nn));
    try {
        dev::memcpy_h2d(Bd, B_host, sizeof(double) * naux * nn * nn, stream);
        TView Bv = make_view(Bd, {naux, nn, nn});
        for (int pat = 0; pat < 16; ++pat) {
            if (!V_[pat]) V_[pat] = static_cast<double*>(dev::dmalloc(sizeof(double) * block_size(pat)));
            auto rng = [&](int pos, int64_t& lo, int64_t& hi) {
                const bool virt = pat >> (3 - pos) & 1;
                lo = virt ? no : 0;
                hi = virt ? nn : no;
            };
            int64_t lo, hi;
            TView Bpr = Bv, Bqs = Bv;
            rng(0, lo, hi); Bpr = slice(Bpr, 1, lo, hi);
            rng(2, lo, hi); Bpr = slice(Bpr, 2, lo, hi);
            rng(1, lo, hi); Bqs = slice(Bqs, 1, lo, hi);
            rng(3, lo, hi); Bqs = slice(Bqs, 2, lo, hi);
            contract(1.0, Bpr, "Qpr", Bqs, "Qqs", 0.0, block_view(V_[pat], pat), "pqrs", "pq");
        }
        dev::stream_sync(stream);
    } catch (...) {
        dev::dfree(Bd);
        throw;
    }
    dev::dfree(Bd);
}

}  // namespace pymes
