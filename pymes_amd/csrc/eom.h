// EOM-CCSD sigma build (pymes/solver/eom_ccsd.py:268-385) on the engine: the u-independent V.T intermediates hoisted once
// per solve (prepare), then H-bar . u for one trial vector or k stacked ones (apply), and the two diagonals
// (eom_ccsd.py:169-266).  The whole-step entry the boundary contract of SURVEY 8(b) names: eom_sigma(ctx, f~, u1, u2, T2).
#pragma once
#include <cstdint>
#include <map>
#include <vector>

#include "engine.h"

namespace pymes {

class EomSigma {
  public:
    // f_host: the T1-dressed Fock matrix [n,n] (host); t2: the CCSD doubles [v,v,o,o] (device; must outlive the object);
    // dressed: read the engine's T1-DRESSED blocks (dress_V) instead of the blocks as uploaded
    EomSigma(Engine& eng, const double* f_host, const double* t2, bool dressed);
    ~EomSigma();
    EomSigma(const EomSigma&) = delete;
    EomSigma& operator=(const EomSigma&) = delete;

    // bit 0: V_abcd = V_badc, 1: T_abij = T_baji, 2: the hole-ladder-shaped terms may run pair-packed, 3: fused pair kernels
    // available for this nocc, 4: the stacked multi-vector build is available (all of the above)
    int flags() const;
    // sigma for k trial vectors: s1[z] [v,o], s2[z] [v,v,o,o] (device, written).  sym[z] != 0: the caller knows that
    // u2[z]_abij = u2[z]_baji; sym == nullptr: tested here (one reduction and a synchronisation per vector)
    void apply(int k, const double* const* u1, const double* const* u2, const int* sym, double* const* s1, double* const* s2);
    bool exchange_symmetric(const double* x, int64_t d0, int64_t d2) const;
    Engine& engine() { return e; }
    void trim();                      // release the pooled temporaries

  private:
    Engine& e;
    const int no, nv;
    const bool dressed;
    const double* T;
    std::vector<double*> owned_;      // hoisted intermediates (engine scratch), returned by the destructor
    double* get(int64_t doubles);
    void put(double* p);
    double* keep(int64_t doubles);
    struct Tmp;                       // RAII temporary from the pool
    // hoisted quantities (names as pymes_amd/solver/eom_ccsd.py round 4, which this replaces)
    double *foo = nullptr, *fov = nullptr, *fvv = nullptr, *fovT = nullptr;
    double *Td = nullptr, *Tx = nullptr, *W1 = nullptr, *Gvv_s = nullptr, *Goo_s = nullptr, *M_C = nullptr, *M_D = nullptr,
           *M1 = nullptr, *Ud = nullptr, *M2 = nullptr, *M12 = nullptr, *MDU = nullptr, *WA = nullptr, *W3 = nullptr, *A3 = nullptr,
           *A4 = nullptr, *A6 = nullptr, *Gvv = nullptr, *Goo = nullptr, *B2 = nullptr, *L = nullptr, *WW = nullptr, *BB = nullptr,
           *Aoo = nullptr, *A346 = nullptr, *TA = nullptr, *LK3 = nullptr, *LK2 = nullptr;
    void general_operands();          // LK3, LK2 (trial vectors without exchange symmetry), on first use
    bool v_sym = false, t_sym = false, hole_sym = false, fused_ok = false, many_ok = false;
    TView V(const char* name) const;
    void singles(const double* u1, const double* u2, double* s1);
    void doubles(const double* u1, const double* u2, bool u2_sym, double* s2, bool defer_ladder = false);
    void general_ladders(int g, const double* const* u2, double* const* s2);
    void stack(int k, const double* const* u1, const double* const* u2, double* const* s1, double* const* s2);
    int stack_limit() const;
};

// eom_ccsd.py:169-198 (get_diag_singles) / :200-266 (get_diag_doubles): d1 [v,o], d2 [v,v,o,o] (device, written); f_host the
// dressed Fock matrix [n,n], t2 [v,v,o,o] on the device, blocks read dressed or as set
void eom_diagonals(Engine& e, const double* f_host, const double* t2, bool dressed, double* d1, double* d2);

}  // namespace pymes
