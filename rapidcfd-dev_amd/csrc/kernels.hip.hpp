// kernels.hip.hpp -- hand-written gfx950 (CDNA4) kernels of the lduMatrix hot path.
//
// Design (DESIGN.md "Kernels"): everything here is HBM-bandwidth bound
// (0.19 flop/byte), so there is no MFMA; the work is organised so that every
// HBM byte is touched once with wide coalesced loads and all irregular
// (gather) accesses happen in LDS:
//   * tile_kernel: one workgroup per tile.  Streams the tile's face
//     coefficients (each symmetric coefficient ONCE) and the tile's psi + halo
//     into LDS, then each wavefront walks 64 rows with a uniform trip count,
//     reading the {slot, other} entries coalesced and gathering from LDS.
//     The row sum is the same fma chain, in the same order, as the reference's
//     matrixMultiplyFunctor (lduMatrixATmul.C:42-138) => bit-identical to the
//     oracle.  Template OP selects Amul/Tmul, sumA, residual, H, H1, the AINV
//     preconditioner (AINVPreconditionerF.H:41-99) and the Jacobi smoother
//     (JacobiSmootherF.H:50-108) -- K1,K2,K3,K6,K8,K18 of SURVEY.md 2.3.
//   * streaming vector kernels with fused reductions (K7,K9-K13): 16-byte
//     loads, fixed block->chunk mapping, wavefront butterfly on ds_swizzle,
//     so every reduction is deterministic run to run.
// Compiled with -ffp-contract=off: every fused multiply-add is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

constexpr int RG = 1024;      // blocks of every streaming/reduction kernel (fixed => deterministic)
static_assert(RG == 1024, "the fold kernels (k_fold_partials / k_fold_final: one 1024-thread workgroup, one slot per thread) assume RG == 1024; 2048 was tried and is NOT a drop-in change");
#ifndef MI_RB
#define MI_RB 512   // 1024 x 512 threads = 32 waves per CU: +2 % PCG iterations/s over 256 (profiles/r02_b_cache_policy_ab.md)
#endif
constexpr int RB = MI_RB;     // threads per block of the streaming kernels

// ---------------------------------------------------------------------------
// wavefront (64 lanes) sum: xor butterfly.  Steps 1..16 stay inside a 32-lane
// half and use ds_swizzle in bit-mask mode (and=0x1f, or=0, xor=m : offset =
// 0x1f | m<<10); the last step crosses the halves with a DPP-free bpermute.
// Every lane ends with the same bits (a+b is commutative).
// ---------------------------------------------------------------------------
template <int XORMASK>
__device__ __forceinline__ double swz_xor(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, 0x1f | (XORMASK << 10));
    hi = __builtin_amdgcn_ds_swizzle(hi, 0x1f | (XORMASK << 10));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v)
{
    v += swz_xor<1>(v);
    v += swz_xor<2>(v);
    v += swz_xor<4>(v);
    v += swz_xor<8>(v);
    v += swz_xor<16>(v);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// block sum for BS threads; result valid in every thread; red = BS/64 doubles of LDS.
template <int BS>
__device__ __forceinline__ double block_sum(double v, double* red)
{
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads(); // protect red from a previous use
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int w = 1; w < BS / 64; ++w) t += red[w];
    return t;
}

// sum of the RG per-block partials, computed redundantly by every block (same
// order everywhere => every block sees the same bits).
__device__ __forceinline__ double sum_partials(const double* __restrict__ partial, double* red)
{
    double v = 0;
#pragma unroll
    for (int k = 0; k < RG / RB; ++k) v += partial[threadIdx.x + k * RB];
    return block_sum<RB>(v, red);
}

// ---------------------------------------------------------------------------
// tile kernel
// ---------------------------------------------------------------------------
enum { OP_AMUL = 0, OP_SUMA = 1, OP_RESIDUAL = 2, OP_H = 3, OP_H1 = 4, OP_AINV = 5, OP_JACOBI = 6,
       // the start of every solver in ONE pass over the coefficients (PCG.C:91-121 and siblings, lduMatrixSolver.C:182-236):
       // y = A x (the Amul's fma chain), y2 = b - y, y3 = lduMatrix::sumA (the sumA pass's chain of additions) -- both chains
       // term by term, so every output bit equals the separate passes'
       OP_PROLOGUE = 7 };

struct TileArgs {
    const int32_t* tileCellStart;
    const int32_t* tileSlotStart;
    const int32_t* tileIfaceSlot0;
    const int32_t* tileHaloStart;
    const int32_t* haloCell;
    const int32_t* tileSliceStart;
    const int32_t* sliceEntryStart;
    const uint32_t* entries;
    const uint32_t* entries16;         // compact form (C16 kernels): two 16-bit entries per word
    const int32_t* sliceEntryStart16;
    const uint32_t* slotBase;          // per tile nc + nh + 1 16-bit slot bases, two per word
    const int32_t* tileSbStart;        // [nTiles+1] in words
    const int32_t* tileList; // nullptr => identity
    int32_t nPos;            // number of tile positions of this launch (== gridDim.x unless the launch is persistent)
    const int32_t* done;     // device-resident solver loops: &PcgState::done, launches past convergence exit at once (else nullptr)
    const double* diag;
    const double* up;
    const double* low;
    // caller-order launch (tile_kernel_perm: mi_amul / mi_tmul / mi_residual / mi_H / mi_sumA / mi_H1 on an addressing that
    // permutes): x, b and y are the CALLER's arrays; perm = e2c (engine cell -> caller cell), haloSrc[h] = caller cell of halo
    // entry h, or -1-k for ext value k of xExt.  Unused by tile_kernel (engine-order vectors: every solver loop).
    const int32_t* perm = nullptr;
    const int32_t* haloSrc = nullptr;
    const double* xExt = nullptr;
    int32_t nCells = 0; // EXTWIN launches (peer.inc: tile_kernel_dist): halo entries >= nCells read xExt[entry - nCells], the rank's halo WINDOW
    // EXTWIN == 2 (peer.inc: tile_kernel_win): the window holds self-validating PAIRS {bits, bits ^ xKey}; a halo entry >= nCells
    // polls its own pair until the two words agree for the key of the current exchange (bounded: xPolls, then *xStatus = 1)
    const unsigned long long* xPairs = nullptr;
    unsigned long long xKey = 0;
    int32_t* xStatus = nullptr;
    long long xPolls = 0;
    const double* x;  // psi (Amul, residual, H, Jacobi) or r (AINV)
    const double* b;  // source (residual, Jacobi)
    const double* rD; // AINV
    double* y;
    double* y2 = nullptr; // OP_PROLOGUE: b - y (rA = source - A psi), or nullptr
    double* y3 = nullptr; // OP_PROLOGUE: sumA, or nullptr (valid from an earlier solve on the same coefficients)
    double* dotPartial; // per-workgroup partial fused into the pass, or nullptr: Amul sum(y*x) (gSumProd), AINV sum(w*r), residual sum|r| (gSumMag)
    double* dotPartial2; // Amul only: per-workgroup partial of sum(b*x) with b = the `b` vector (GAMG scale: gSumProd(source, field)), or nullptr
    double omega;
    int32_t offLow, offX, offRD, offSB; // LDS offsets in doubles
    int32_t flags;               // bit0: non-temporal coefficient loads, bit1: non-temporal entry loads, bit2: nt result stores, bit3: nt diagonal loads (Amul)
    // XMODE launches (tile_kernel_fx; GAMG, gamg_engine.inc): the operand x is FORMED while it is staged instead of read --
    //   1: x[c] = fxCoarse[fxMap[c]]                                    (GAMGAgglomeration::prolongField, GAMGAgglomerationTemplates.C:273-308),
    //      written to fxOut for the own cells when fxOut != nullptr
    //   2: x[c] = sf*fxField[c] + (fxSrc[c] - sf*fxAcf[c])/diag[c],  sf = num / stabilise(den)   (GAMGSolver::scale, GAMGSolverScale.C:59-171),
    //      {num, den} = fxScal[0..1]
    //   3: x[c] = fxAdd[c] + the value of 2                               (psi += finestCorrection, GAMGSolverSolve.C:130-138)
    // -- the prolongation kernel and the scaling pass of a level disappear into the tile pass that consumes their result, same arithmetic.
    const int32_t* fxMap = nullptr; const double* fxCoarse = nullptr; double* fxOut = nullptr;
    const double *fxField = nullptr, *fxAcf = nullptr, *fxSrc = nullptr, *fxScal = nullptr, *fxAdd = nullptr;
    // Amul with both fused sums (dotPartial, dotPartial2): the LAST workgroup to finish adds the per-tile partials in the order
    // k_fold_partials2 + sum_partials add them and stores {sum of dotPartial2, sum of dotPartial} = {num, den} to foldOut
    unsigned int* foldCounter = nullptr; double* foldOut = nullptr;
};
constexpr double FX_VSMALL = 1e-300;   // SP_VSMALL (below)

// cooperative global -> LDS staging, 4 loads in flight per lane
template <bool NT, class T>
__device__ __forceinline__ T ldg(const T* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ double2 ldg2(const double2* p, bool nt)
{
    if (!nt) return *p;
    const double* q = reinterpret_cast<const double*>(p);
    typedef double dvec2 __attribute__((ext_vector_type(2)));
    const dvec2 v = __builtin_nontemporal_load(reinterpret_cast<const dvec2*>(q));
    return make_double2(v.x, v.y);
}
template <int BS, class T>
__device__ __forceinline__ void stage_copy(const T* __restrict__ src, T* __restrict__ dst, int n, int tid)
{
    int k = tid;
    for (; k + 3 * BS < n; k += 4 * BS) {
        const T v0 = src[k], v1 = src[k + BS], v2 = src[k + 2 * BS], v3 = src[k + 3 * BS];
        dst[k] = v0; dst[k + BS] = v1; dst[k + 2 * BS] = v2; dst[k + 3 * BS] = v3;
    }
    for (; k < n; k += BS) dst[k] = src[k];
}
// streamed-once data (coefficients): non-temporal loads so they do not displace psi/halo lines in L2
template <int BS>
__device__ __forceinline__ void stage_copy_nt(const double2* __restrict__ src, double2* __restrict__ dst, int n, int tid, bool nt)
{
    int k = tid;
    for (; k + 3 * BS < n; k += 4 * BS) {
        const double2 v0 = ldg2(src + k, nt), v1 = ldg2(src + k + BS, nt), v2 = ldg2(src + k + 2 * BS, nt), v3 = ldg2(src + k + 3 * BS, nt);
        dst[k] = v0; dst[k + BS] = v1; dst[k + 2 * BS] = v2; dst[k + 3 * BS] = v3;
    }
    for (; k < n; k += BS) dst[k] = ldg2(src + k, nt);
}
template <int BS>
__device__ __forceinline__ void stage_gather(const double* __restrict__ x, const int32_t* __restrict__ idx,
                                             double* __restrict__ dst, int n, int tid)
{
    int k = tid;
    for (; k + BS < n; k += 2 * BS) {
        const int i0 = idx[k], i1 = idx[k + BS];
        const double v0 = x[i0], v1 = x[i1];
        dst[k] = v0; dst[k + BS] = v1;
    }
    for (; k < n; k += BS) dst[k] = x[idx[k]];
}

// Direct-to-LDS staging (global_load_lds_dwordx4): every lane fetches 16 bytes from its own global address and the wave's
// 1 KiB lands contiguously in LDS at a wave-uniform base -- no VGPR round trip, one instruction per KiB per wave.  Measured
// in the standalone harness on one box: 155.7 -> 147.8 us for the symmetric Amul (profiles/r01_t_direct_to_lds.md).
typedef __attribute__((address_space(1))) const void* mi_gptr_t;
typedef __attribute__((address_space(3))) void* mi_lptr_t;
template <int BS, bool NT = false>
__device__ __forceinline__ void stage_dma16(const double2* __restrict__ src, double2* __restrict__ dst, int n2, int wave, int lane)
{
    for (int base = wave * 64; base < n2; base += BS)
        if (base + lane < n2) __builtin_amdgcn_global_load_lds((mi_gptr_t)(src + base + lane), (mi_lptr_t)(dst + base), 16, 0, NT ? 2 : 0); // aux 2 = nt
}
// n doubles (8-byte aligned source): pairs through the DMA path, an odd last element by hand
template <int BS>
__device__ __forceinline__ void stage_dma8(const double* __restrict__ src, double* __restrict__ dst, int n, int tid)
{
    stage_dma16<BS>(reinterpret_cast<const double2*>(src), reinterpret_cast<double2*>(dst), n >> 1, tid >> 6, tid & 63);
    if ((n & 1) && tid == 0) dst[n - 1] = src[n - 1];
}

// one tile: position p of the launch (p indexes tileList / dotPartial)
template <int BS>
__device__ __forceinline__ void fold_two_partials(const double* pA, const double* pB, int n, double* lds, double* out);
template <int OP, bool ASYM, bool TRANS, int BS, bool C16, bool PERM = false, int EXTWIN = 0, int XMODE = 0>
__device__ __forceinline__ void tile_body(const TileArgs& a, const int p, double* __restrict__ smem)
{
    double* cU = smem;
    double* cL = smem + a.offLow;
    double* xs = smem + a.offX;
    double* rDs = smem + a.offRD;
    const int t = a.tileList ? a.tileList[p] : p;

    const int tid = threadIdx.x;
    const int c0 = a.tileCellStart[t], nc = a.tileCellStart[t + 1] - c0;
    const int s0 = a.tileSlotStart[t], ns = a.tileSlotStart[t + 1] - s0; // even
    const int h0 = a.tileHaloStart[t], nh = a.tileHaloStart[t + 1] - h0;
    const int ifs0 = (OP == OP_JACOBI || OP == OP_AINV || OP == OP_H || OP == OP_H1 || (C16 && ASYM)) ? a.tileIfaceSlot0[t] : 0; // first interface slot of the tile
    uint16_t* sb = reinterpret_cast<uint16_t*>(smem + a.offSB);
    constexpr bool NEEDX = (OP != OP_SUMA && OP != OP_H1);
    static_assert(!PERM || OP == OP_AMUL || OP == OP_RESIDUAL || OP == OP_H || OP == OP_SUMA || OP == OP_H1, "caller-order form: ops with a caller-order entry point only");
    static_assert(OP != OP_PROLOGUE || XMODE == 0, "the solver prologue reads its operand");

    // ---- stage: coefficients (16-byte coalesced), psi, halo -------------------
    // All global loads of a phase are issued before the first LDS store (4-deep
    // unroll) so that every wave keeps several KiB in flight.
    // measured per variant on one box (profiles/r01_t_direct_to_lds.md): faster everywhere except the asymmetric AINV pass
    // (four LDS arrays, one 1024-thread workgroup per CU), which keeps the register path
    constexpr bool DMA = !(OP == OP_AINV && ASYM);
    if (DMA) {
    if (a.flags & 1) { // coefficients are read once per launch: non-temporal
        stage_dma16<BS, true>(reinterpret_cast<const double2*>(a.up + s0), reinterpret_cast<double2*>(cU), ns >> 1, tid >> 6, tid & 63);
        if (ASYM) stage_dma16<BS, true>(reinterpret_cast<const double2*>(a.low + s0), reinterpret_cast<double2*>(cL), ns >> 1, tid >> 6, tid & 63);
    } else {
        stage_dma16<BS>(reinterpret_cast<const double2*>(a.up + s0), reinterpret_cast<double2*>(cU), ns >> 1, tid >> 6, tid & 63);
        if (ASYM) stage_dma16<BS>(reinterpret_cast<const double2*>(a.low + s0), reinterpret_cast<double2*>(cL), ns >> 1, tid >> 6, tid & 63);
    }
    if (NEEDX) {
        if (XMODE != 0) { // GAMG: prolongation / correction scaling folded into the staging (see TileArgs)
            double sf = 0.0;
            if (XMODE >= 2) { const double num = a.fxScal[0], den = a.fxScal[1]; sf = num / (den >= 0 ? den + FX_VSMALL : den - FX_VSMALL); }
            auto fx = [&](int c) -> double {
                if (XMODE == 1) return a.fxCoarse[a.fxMap[c]];
                const double v = fma(sf, a.fxField[c], fma(-sf, a.fxAcf[c], a.fxSrc[c]) / a.diag[c]);
                return XMODE == 3 ? a.fxAdd[c] + v : v;
            };
            if (XMODE == 1) {
                // own and halo cells in ONE index space, four per lane at a time: all cell indices, then all parents, then all coarse values --
                // three rounds of independent loads instead of a dependent chain per cell (the staging is latency-bound: nothing else is in
                // flight before the barrier)
                for (int k0 = 0; k0 < nc + nh; k0 += 4 * BS) {
                    int cc[4]; double vv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int k = k0 + j * BS + tid; cc[j] = k < nc ? c0 + k : (k < nc + nh ? a.haloCell[h0 + k - nc] : -1); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) cc[j] = cc[j] >= 0 ? a.fxMap[cc[j]] : -1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) vv[j] = cc[j] >= 0 ? a.fxCoarse[cc[j]] : 0.0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int k = k0 + j * BS + tid; if (k < nc + nh) { xs[k] = vv[j]; if (a.fxOut && k < nc) a.fxOut[c0 + k] = vv[j]; } }
                }
            } else {
            for (int k = tid; k < nc; k += BS) xs[k] = fx(c0 + k);
            for (int k = tid; k < nh; k += BS) xs[nc + k] = fx(a.haloCell[h0 + k]);
            }
        } else if (PERM) { // the permutation of the caller-order entry points, folded into the staging
            stage_gather<BS>(a.x, a.perm + c0, xs, nc, tid);
            for (int k = tid; k < nh; k += BS) { const int src = a.haloSrc[h0 + k]; xs[nc + k] = src >= 0 ? a.x[src] : a.xExt[-1 - src]; }
        } else {
        stage_dma8<BS>(a.x + c0, xs, nc, tid);
        if (EXTWIN == 2) { // ... as self-validating pairs: no flag, no fence between value and flag on the writer's side -- the value IS the flag
            for (int k = tid; k < nh; k += BS) {
                const int idx = a.haloCell[h0 + k];
                double v;
                if (idx < a.nCells) v = a.x[idx];
                else {
                    const unsigned long long* q = a.xPairs + 2 * (size_t)(idx - a.nCells);
                    unsigned long long bits = 0;
                    long long polls = 0;
                    for (;;) {
                        bits = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        const unsigned long long chk = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if ((bits ^ chk) == a.xKey) break;
                        // a neighbour that never arrives must not hang the device: the status word is read and written with atomics
                        // (every lane sees another lane's fault; a plain load could be hoisted out of the loop), and the lane that
                        // gives up stages a NaN, so the fault reaches the sums and the caller's peer_check
                        if (++polls > a.xPolls || __hip_atomic_load(a.xStatus, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            __hip_atomic_store(a.xStatus, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            bits = 0x7ff8000000000000ull;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    v = __longlong_as_double((long long)bits);
                }
                xs[nc + k] = v;
            }
        } else if (EXTWIN) { // neighbour ranks' values come straight from this rank's halo window (written by the peers, system scope)
            for (int k = tid; k < nh; k += BS) {
                const int idx = a.haloCell[h0 + k];
                xs[nc + k] = idx < a.nCells ? a.x[idx] : __hip_atomic_load(a.xExt + (idx - a.nCells), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        } else
        stage_gather<BS>(a.x, a.haloCell + h0, xs + nc, nh, tid);
        }
        if (OP == OP_AINV) {
            stage_dma8<BS>(a.rD + c0, rDs, nc, tid);
            stage_gather<BS>(a.rD, a.haloCell + h0, rDs + nc, nh, tid);
        }
    }
    } else {
        stage_copy_nt<BS>(reinterpret_cast<const double2*>(a.up + s0), reinterpret_cast<double2*>(cU), ns >> 1, tid, false);
        stage_copy_nt<BS>(reinterpret_cast<const double2*>(a.low + s0), reinterpret_cast<double2*>(cL), ns >> 1, tid, false);
        stage_copy<BS>(a.x + c0, xs, nc, tid);
        stage_gather<BS>(a.x, a.haloCell + h0, xs + nc, nh, tid);
        stage_copy<BS>(a.rD + c0, rDs, nc, tid);
        stage_gather<BS>(a.rD, a.haloCell + h0, rDs + nc, nh, tid);
    }
    if (C16) { // slot bases of the tile's cells, halo cells and the pad cell (whose x is 0 and whose slot is the zero slot)
        const int w0 = a.tileSbStart[t], nw = a.tileSbStart[t + 1] - w0;
#pragma unroll 2
        for (int k = tid; k < nw; k += BS) reinterpret_cast<uint32_t*>(sb)[k] = a.slotBase[w0 + k];
        if (tid == 0 && NEEDX) { xs[nc + nh] = 0.0; if (OP == OP_AINV) rDs[nc + nh] = 0.0; }
    }

    // ---- rows: one wavefront per 64-row slice, uniform trip count -------------
    // The {slot, other} entries of a slice are fetched into registers one slice
    // ahead (the first slice before the staging barrier), so the compute loop
    // only touches LDS.
    const int sl0 = a.tileSliceStart[t], nsl = a.tileSliceStart[t + 1] - sl0;
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int NW = BS / 64;
    constexpr int PRE = C16 ? 4 : 8; // words held in registers: 8 entries either way
    const uint32_t pad16 = (uint32_t)(nc + nh) | 0x8000u;
    const uint32_t padEnt = C16 ? (pad16 | (pad16 << 16)) : (uint32_t)(ns - 1) << 16; // last slot of the segment is always 0.0
    const uint32_t* const entBase = C16 ? a.entries16 : a.entries;
    const int32_t* const sesBase = C16 ? a.sliceEntryStart16 : a.sliceEntryStart;
    uint32_t ecur[PRE];
    int wcur = 0, e0cur = 0;
    auto fetch = [&](int s, uint32_t (&e)[PRE], int& e0, int& width) {
        e0 = __builtin_amdgcn_readfirstlane(sesBase[sl0 + s]);
        const int e1 = __builtin_amdgcn_readfirstlane(sesBase[sl0 + s + 1]);
        width = (e1 - e0) >> 6;
        const uint32_t* ent = entBase + e0 + lane;
#pragma unroll
        for (int j = 0; j < PRE; ++j) e[j] = (j < width) ? ((a.flags & 2) ? __builtin_nontemporal_load(ent + j * 64) : ent[j * 64]) : padEnt;
    };
    if (wave < nsl) fetch(wave, ecur, e0cur, wcur);
    else {
#pragma unroll
        for (int j = 0; j < PRE; ++j) ecur[j] = padEnt;
    }
    __syncthreads();

    double dot = 0.0, dot2 = 0.0;
    for (int s = wave; s < nsl; s += NW) {
        uint32_t enext[PRE];
        int wnext = 0, e0next = 0;
        if (s + NW < nsl) fetch(s + NW, enext, e0next, wnext);
        else {
#pragma unroll
            for (int j = 0; j < PRE; ++j) enext[j] = padEnt;
        }
        const int i = s * 64 + lane;
        const bool live = i < nc;
        const int gi = c0 + (live ? i : 0);
        const int go = PERM ? a.perm[gi] : gi; // index into the caller's arrays (b, y)
        const double xi = (NEEDX && live) ? xs[i] : 0.0;
        double acc, accI = 0.0;
        if (OP == OP_JACOBI) accI = a.b[gi];
        if (OP == OP_PROLOGUE) { accI = a.diag[gi]; acc = accI * xi; }   // accI: the row's sumA
        else if (OP == OP_AMUL) acc = ((a.flags & 8) ? __builtin_nontemporal_load(a.diag + gi) : a.diag[gi]) * xi;
        else if (OP == OP_SUMA) acc = a.diag[gi];
        else if (OP == OP_RESIDUAL) acc = a.b[go] - a.diag[gi] * xi;
        else acc = 0.0;
        auto apply = [&](const int o, const int sl, const bool lowerSide) {
            double c;
            if (ASYM) c = (lowerSide != TRANS) ? cL[sl] : cU[sl];
            else c = cU[sl];
            if (OP == OP_JACOBI) { // coupled patches go to bPrime, faces to the row sum (JacobiSmoother.C:75-93, JacobiSmootherF.H)
                if (sl >= ifs0) accI = fma(-c, xs[o], accI); else acc = fma(c, xs[o], acc);
            } else if (OP == OP_AMUL) acc = fma(c, xs[o], acc);
            else if (OP == OP_PROLOGUE) { acc = fma(c, xs[o], acc); accI += c; }
            else if (OP == OP_SUMA) acc += c;
            else if (OP == OP_RESIDUAL) acc = fma(-c, xs[o], acc);
            else if (OP == OP_H) { if (sl < ifs0) acc = fma(-c, xs[o], acc); }   // lduMatrix::H / H1 are face sums only
            else if (OP == OP_H1) { if (sl < ifs0) acc -= c; }                     // (lduMatrixOperations.C:130-154, lduMatrixATmul.C:533-554)
            else if (OP == OP_AINV) { if (sl < ifs0) acc = fma(c * rDs[o], xs[o], acc); } // faces only (AINVPreconditioner.C)
        };
        auto accumulate = [&](uint32_t en) { apply(en & 0xFFFFu, (en >> 16) & 0x7FFFu, (en >> 31) != 0u); };
        // compact entry j of the row: rule 0 -> the row's own run of slots, rule 1 -> the other cell's run (+k); the
        // coefficient is the lower one for rule-1 faces (interfaces, slot >= ifs0, always take the upper array)
        const int sbi = C16 ? (int)sb[live ? i : 0] : 0;
        auto accumulate16 = [&](uint32_t e16, int j) {
            const int o = e16 & 0xFFFu;
            const bool rule = (e16 & 0x8000u) != 0u;
            const int sl = rule ? (int)sb[o] + (int)((e16 >> 12) & 7u) : sbi + j;
            apply(o, sl, ASYM ? (rule && sl < ifs0) : false);
        };
        if (C16) {
#pragma unroll
            for (int j = 0; j < PRE; ++j) if (j < wcur) { accumulate16(ecur[j] & 0xFFFFu, 2 * j); accumulate16(ecur[j] >> 16, 2 * j + 1); }
            if (wcur > PRE) {
                const uint32_t* ent = entBase + e0cur + lane;
                for (int j = PRE; j < wcur; ++j) { const uint32_t w = ent[j * 64]; accumulate16(w & 0xFFFFu, 2 * j); accumulate16(w >> 16, 2 * j + 1); }
            }
        } else {
#pragma unroll
        for (int j = 0; j < PRE; ++j) if (j < wcur) accumulate(ecur[j]);
        if (wcur > PRE) {
            const uint32_t* ent = a.entries + e0cur + lane;
            for (int j = PRE; j < wcur; ++j) accumulate(ent[j * 64]);
        }
        }
        if (live) {
            if (OP == OP_AINV) { const double w = rDs[i] * (xi - acc); a.y[gi] = w; dot = fma(w, xi, dot); } // + fused gSumProd(wA, rA), PCG.C:139-142
            else if (OP == OP_JACOBI) {
                const double rD = 1.0 / a.diag[gi];
                const double extra = (1 - a.omega) * xi + a.omega * rD * accI;
                a.y[gi] = extra - a.omega * rD * acc;
            } else if (OP == OP_PROLOGUE) {
                a.y[gi] = acc;
                if (a.y2) a.y2[gi] = a.b[gi] - acc;
                if (a.y3) a.y3[gi] = accI;
            } else if (a.flags & 4) __builtin_nontemporal_store(acc, a.y + go);
            else a.y[go] = acc;
            if (OP == OP_AMUL) { dot = fma(acc, xi, dot); if (a.dotPartial2) dot2 = fma(a.b[gi], xi, dot2); }
            if (OP == OP_RESIDUAL) dot += fabs(acc);
        }
#pragma unroll
        for (int j = 0; j < PRE; ++j) ecur[j] = enext[j];
        wcur = wnext; e0cur = e0next;
    }
    if ((OP == OP_AMUL || OP == OP_AINV || OP == OP_RESIDUAL) && a.dotPartial) { // fused gSumProd(wA, pA), PCG.C:166 / gSumProd(wA, rA), PCG.C:142 / gSumMag(rA)
        __shared__ double red[BS / 64];
        const double tsum = block_sum<BS>(dot, red);
        if (XMODE != 0 && a.foldOut) { if (tid == 0) __hip_atomic_store(a.dotPartial + p, tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else if (tid == 0) a.dotPartial[p] = tsum;
        if (OP == OP_AMUL && a.dotPartial2) {
            const double t2 = block_sum<BS>(dot2, red);
            if (XMODE != 0 && a.foldOut) { if (tid == 0) __hip_atomic_store(a.dotPartial2 + p, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else if (tid == 0) a.dotPartial2[p] = t2;
        }
        if (XMODE != 0 && OP == OP_AMUL && a.foldOut) {   // the last workgroup to arrive folds the partials of all tiles
            __shared__ unsigned int lastOne;
            // no agent-scope fence here: an agent-scope acquire / release invalidates / writes back the whole L2 of the XCD under the tiles
            // that are still computing (first version: 3.7 x slower).  The partials were written with agent-scope (write-through) stores and
            // are read with agent-scope loads; the arrival only has to be ORDERED behind this workgroup's two stores (persist.inc: grid_barrier)
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);
                lastOne = (__hip_atomic_fetch_add(a.foldCounter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.nPos - 1u) ? 1u : 0u;
            }
            __syncthreads();
            if (lastOne) {
                fold_two_partials<BS>(a.dotPartial2, a.dotPartial, a.nPos, smem, a.foldOut);
                if (tid == 0) __hip_atomic_store(a.foldCounter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// XCD-aware mapping: hardware places block b on XCD b%8; every XCD gets a contiguous run of tile positions so that
// neighbouring tiles (which share halo cells) hit the same L2.  Speed only, never correctness.
//   gridDim.x == nPos : one workgroup per tile.
//   gridDim.x <  nPos : PERSISTENT launch -- as many workgroups as the chip holds at once (a multiple of 8), each walks
//                       a contiguous run of its XCD's tiles; saves the dispatch/teardown of ~10 short workgroups per slot.
// Register budget: 512- and 1024-thread workgroups are launched where four / two of them are meant to share a CU (32 waves,
// 8 per SIMD): that needs <= 64 VGPRs per lane, which the compiler is told here (without it the symmetric Amul takes 66 and
// the symmetric Jacobi sweep 75, and a CU holds three workgroups instead of four).
// (the opt-in compact asymmetric form would spill two registers under that budget and is left alone)
template <int OP, bool ASYM, bool TRANS, int BS, bool C16>
__global__ __launch_bounds__(BS, (BS >= 512 && !(C16 && ASYM) && OP != OP_PROLOGUE) ? 8 : 1) void tile_kernel(const TileArgs a)   // (the prologue runs once per solve and carries two accumulators: no register cap)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (a.done && *a.done) return;
    const int b = blockIdx.x, G = gridDim.x, nT = a.nPos;
    if (G >= nT) {
        const int per = G >> 3;
        tile_body<OP, ASYM, TRANS, BS, C16>(a, (b < (per << 3)) ? (b & 7) * per + (b >> 3) : b, smem);
        return;
    }
    const int per = G >> 3, x = b & 7, j = b >> 3;
    const int x0 = (int)((long long)x * nT / 8), x1 = (int)((long long)(x + 1) * nT / 8);
    const int p0 = x0 + (int)((long long)j * (x1 - x0) / per), p1 = x0 + (int)((long long)(j + 1) * (x1 - x0) / per);
    for (int p = p0; p < p1; ++p) {
        tile_body<OP, ASYM, TRANS, BS, C16>(a, p, smem);
        __syncthreads(); // every wave is done with the LDS image before the next tile is staged
    }
}

// the same tile pass on the CALLER's arrays: x gathered through e2c while it is staged, y (and b) addressed through e2c in the
// row loop -- the two permutation passes of the caller-order entry points folded into the kernel.  Separate instantiation, so
// that the engine-order kernel above keeps its registers (a run-time switch cost it two spills).
template <int OP, bool ASYM, bool TRANS, int BS>
__global__ __launch_bounds__(BS) void tile_kernel_perm(const TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x, per = gridDim.x >> 3;
    tile_body<OP, ASYM, TRANS, BS, false, true>(a, (b < (per << 3)) ? (b & 7) * per + (b >> 3) : b, smem);
}

// tile pass whose operand is formed while it is staged (XMODE, see TileArgs): GAMG's prolongation and correction scaling fused
// into the Amul / Jacobi pass that consumes them.  Separate instantiations (the solver loops' kernels keep their registers).
template <int OP, bool ASYM, int BS, int XMODE>
__global__ __launch_bounds__(BS, (BS >= 512 && !ASYM) ? 8 : 1) void tile_kernel_fx(const TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x, per = gridDim.x >> 3;
    tile_body<OP, ASYM, false, BS, false, false, 0, XMODE>(a, (b < (per << 3)) ? (b & 7) * per + (b >> 3) : b, smem);
}
// {out[0], out[1]} = the sums of pA[0..n) and pB[0..n) in exactly the order the two-launch path adds them: k_fold_partials2 (only
// when n > RG: slot t = sum of entries t, t + 1024, ...) and then sum_partials on RB threads (thread t: slots t, t + RB, ...;
// wavefront butterflies; the wavefronts' results in order) -- emulated on a workgroup of BS threads; lds: 2 * (RG + RB/64) doubles.
// Entries are read with agent-scope loads (other workgroups of the same launch wrote them with agent-scope stores).
template <int BS>
__device__ __forceinline__ void fold_two_partials(const double* pA, const double* pB, int n, double* lds, double* out)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* lA = lds; double* lB = lds + RG; double* rA = lds + 2 * RG; double* rB = rA + RB / 64;
    __syncthreads();
    for (int t = tid; t < RG; t += BS) {               // both arrays in one sweep: the loads of a slot are independent
        double va = 0.0, vb = 0.0;
        if (n > RG) {
            for (int k = t; k < n; k += 1024) {
                va += __hip_atomic_load(pA + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vb += __hip_atomic_load(pB + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (t < n) {                            // slots >= n were zeroed once and are never written
            va = __hip_atomic_load(pA + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            vb = __hip_atomic_load(pB + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lA[t] = va; lB[t] = vb;
    }
    __syncthreads();
    for (int vw = wave; vw < RB / 64; vw += BS / 64) {
        double va = 0.0, vb = 0.0;
#pragma unroll
        for (int k = 0; k < RG / RB; ++k) { va += lA[vw * 64 + lane + k * RB]; vb += lB[vw * 64 + lane + k * RB]; }
        va = wave_sum(va); vb = wave_sum(vb);
        if (lane == 0) { rA[vw] = va; rB[vw] = vb; }
    }
    __syncthreads();
    if (tid == 0) {
        double ta = rA[0], tb = rB[0];
#pragma unroll
        for (int w = 1; w < RB / 64; ++w) { ta += rA[w]; tb += rB[w]; }
        out[0] = ta; out[1] = tb;
    }
}

// ---------------------------------------------------------------------------
// multi-vector tile pass (round 3): ONE staging of a tile's coefficients serves 2*NRHS operand vectors -- NRHS of them through
// the matrix as bound, NRHS through its transpose.  That is PBiCG's pair  wA = A pA, wT = A^T pT  (PBiCG.C:177-181) and its
// preconditioner pair  precondition(wA, rA), preconditionT(wT, rT)  (PBiCG.C:149-153) in one pass, and for NRHS = 3 the three
// components of fvMatrix<vector>::solveSegregated (fvMatrixSolve.C:148-206), which the reference solves one after the other,
// each re-reading the same upper / lower / addressing: bytes per row 52.6 (both triangles) + 24 (row entries) + 28.8 per
// operand instead of (52.6 + 24 + 28.8) per operand.  Per operand the fma chain is the single-vector kernel's, term by term,
// so every output bit equals the single-vector pass (and the oracle).  Asymmetric matrices (the bi-conjugate solvers' case;
// a symmetric matrix passes its upper array for both triangles).
// ---------------------------------------------------------------------------
template <int NRHS>
struct MultiVec {
    const double* x[2 * NRHS];     // operands: [0, NRHS) plain, [NRHS, 2 NRHS) through the transpose
    double* y[2 * NRHS];           // (OP_SUMA: y[c], c < NRHS, receives sumA of component c)
    const int32_t* done[NRHS];     // &PcgState::done of component c (or nullptr): a finished component costs nothing
    // fvMatrix<vector>::solveSegregated adds the boundary contribution to the diagonal PER COMPONENT (addBoundaryDiag(diag, cmpt),
    // fvMatrixSolve.C:171-176): the components share upper / lower but may differ in the diagonal
    const double* diag[NRHS];      // engine order
    const double* rD[NRHS];        // 1 / diag (AINV); SRD: all equal, staged once
    // fused into the pass (all optional): per-workgroup partials of sum_i y[c][i] * x[NRHS + c][i] -- PBiCG's wA.rT (PBiCG.C:155,
    // out of the preconditioner pass) and wA.pT (PBiCG.C:183, out of the Amul/Tmul pass); and for the solver prologue
    // y2[v] = b[c] - y[v] (rA = source - A psi, rT = source - A^T psi, PBiCG.C:110-128) with y[v] itself optional
    double* dotPartial[NRHS];
    const double* b[NRHS];
    double* y2[2 * NRHS];
    double* sumA[NRHS];            // OP_PROLOGUE: lduMatrix::sumA with component c's diagonal (or nullptr) -- the sumA pass's additions in its order
};

template <int OP, int NRHS, bool SRD, int BS>
__global__ __launch_bounds__(BS) void tile_kernel_multi(const TileArgs a, const MultiVec<NRHS> V)
{
    static_assert(OP == OP_AMUL || OP == OP_AINV || OP == OP_SUMA || OP == OP_PROLOGUE, "multi-vector form: Amul/Tmul pairs (OP_PROLOGUE: + sumA per diagonal in the same pass), AINV / AINV^T pairs, sumA per diagonal");
    constexpr bool MUL = (OP == OP_AMUL || OP == OP_PROLOGUE);
    constexpr int NV = (OP == OP_SUMA) ? NRHS : 2 * NRHS;
    constexpr int NRD = SRD ? 1 : NRHS;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // ONE workgroup per CU (the image of 2 NRHS operands + both triangles takes 134-148 KB of LDS): nothing else hides a memory
    // round trip, so the staging is written as THREE dependent rounds, not one per array (round 4 had ~17: the `done` flags one
    // after the other, then per operand  index -> wait -> value -> wait  behind the coefficient DMA, then the diagonal after the
    // barrier; 19.6 us per tile, 0.28-0.32 of the HBM roofline -- profiles/r04_pbicg_rocprof_summary.md):
    //   1. the components' `done` flags, together;
    //   2. the tile's halo indices (the same for every operand), the first slice's entries;
    //   3. everything else at once: coefficient and own-cell DMA, every operand's halo values, the first slice's diagonals.
    bool act[NRHS];
    bool any = false;
    {
        int dn[NRHS];
#pragma unroll
        for (int c = 0; c < NRHS; ++c) dn[c] = *(V.done[c] ? V.done[c] : &a.tileCellStart[0]);   // (any readable word when there is no flag)
#pragma unroll
        for (int c = 0; c < NRHS; ++c) { act[c] = !(V.done[c] && dn[c]); any = any || act[c]; }
    }
    if (!any) return;
    const int b = blockIdx.x, per = gridDim.x >> 3;
    const int t = (b < (per << 3)) ? (b & 7) * per + (b >> 3) : b;     // XCD-aware, as tile_kernel
    double* cU = smem;
    double* cL = smem + a.offLow;
    double* xs = smem + a.offX;             // NV arrays of xlen doubles
    double* rDs = smem + a.offRD;           // NRD arrays of xlen doubles
    const int xlen = a.offSB;               // (re-used field: doubles per staged operand)
    const int tid = threadIdx.x;
    const int c0 = a.tileCellStart[t], nc = a.tileCellStart[t + 1] - c0;
    const int s0 = a.tileSlotStart[t], ns = a.tileSlotStart[t + 1] - s0;
    const int h0 = a.tileHaloStart[t], nh = a.tileHaloStart[t + 1] - h0;
    const int ifs0 = (OP == OP_AINV) ? a.tileIfaceSlot0[t] : 0;
    const int sl0 = a.tileSliceStart[t], nsl = a.tileSliceStart[t + 1] - sl0;
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int NW = BS / 64;
    constexpr int PRE = 8;
    constexpr int HK = 2;                   // halo values through registers: HK per lane (more halo cells than HK * BS: the loop below)
    const uint32_t padEnt = (uint32_t)(ns - 1) << 16;   // last slot of the segment is always 0.0
    uint32_t ecur[PRE];
    int wcur = 0, e0cur = 0;
    auto fetch = [&](int s, uint32_t (&e)[PRE], int& e0, int& width) {
        e0 = __builtin_amdgcn_readfirstlane(a.sliceEntryStart[sl0 + s]);
        const int e1 = __builtin_amdgcn_readfirstlane(a.sliceEntryStart[sl0 + s + 1]);
        width = (e1 - e0) >> 6;
        const uint32_t* ent = a.entries + e0 + lane;
#pragma unroll
        for (int j = 0; j < PRE; ++j) e[j] = (j < width) ? ent[j * 64] : padEnt;
    };
    // round 2
    int hidx[HK];
#pragma unroll
    for (int k = 0; k < HK; ++k) hidx[k] = (OP != OP_SUMA && tid + k * BS < nh) ? a.haloCell[h0 + tid + k * BS] : -1;
    if (wave < nsl) fetch(wave, ecur, e0cur, wcur);
    else {
#pragma unroll
        for (int j = 0; j < PRE; ++j) ecur[j] = padEnt;
    }
    asm volatile("" ::: "memory");          // (round 2 is issued before round 3, not sunk to its uses)
    // round 3
    double hv[(OP == OP_SUMA) ? 1 : NV][HK], hr[NRD][HK];
    if (OP != OP_SUMA) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int k = 0; k < HK; ++k) hv[v][k] = (hidx[k] >= 0 && act[v % NRHS]) ? V.x[v][hidx[k]] : 0.0;
    }
    if (OP == OP_AINV) {
#pragma unroll
        for (int c = 0; c < NRD; ++c)
#pragma unroll
            for (int k = 0; k < HK; ++k) hr[c][k] = hidx[k] >= 0 ? V.rD[c][hidx[k]] : 0.0;
    }
    double dg0[NRHS];                       // the first slice's diagonals (every wave has exactly one slice when nsl <= NW)
    {
        const int i = wave * 64 + lane;
        const int gi = c0 + ((wave < nsl && i < nc) ? i : 0);
#pragma unroll
        for (int c = 0; c < NRHS; ++c) dg0[c] = (OP == OP_AINV) ? 0.0 : V.diag[c][gi];
    }
    stage_dma16<BS, true>(reinterpret_cast<const double2*>(a.up + s0), reinterpret_cast<double2*>(cU), ns >> 1, tid >> 6, tid & 63);
    stage_dma16<BS, true>(reinterpret_cast<const double2*>(a.low + s0), reinterpret_cast<double2*>(cL), ns >> 1, tid >> 6, tid & 63);
    if (OP != OP_SUMA) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (!act[v % NRHS]) continue;
            stage_dma8<BS>(V.x[v] + c0, xs + v * xlen, nc, tid);
        }
    }
    if (OP == OP_AINV) {
#pragma unroll
        for (int c = 0; c < NRD; ++c) stage_dma8<BS>(V.rD[c] + c0, rDs + c * xlen, nc, tid);
    }
    // every load of round 3 is in flight before the first wait: the LDS writes of the gathered values (which wait for ALL
    // outstanding loads -- one counter) must not be scheduled above the DMA issue
    asm volatile("" ::: "memory");
    if (OP != OP_SUMA) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int k = 0; k < HK; ++k) if (hidx[k] >= 0 && act[v % NRHS]) xs[v * xlen + nc + tid + k * BS] = hv[v][k];
        if (nh > HK * BS) {
#pragma unroll
            for (int v = 0; v < NV; ++v) if (act[v % NRHS]) stage_gather<BS>(V.x[v], a.haloCell + h0 + HK * BS, xs + v * xlen + nc + HK * BS, nh - HK * BS, tid);
        }
    }
    if (OP == OP_AINV) {
#pragma unroll
        for (int c = 0; c < NRD; ++c)
#pragma unroll
            for (int k = 0; k < HK; ++k) if (hidx[k] >= 0) rDs[c * xlen + nc + tid + k * BS] = hr[c][k];
        if (nh > HK * BS) {
#pragma unroll
            for (int c = 0; c < NRD; ++c) stage_gather<BS>(V.rD[c], a.haloCell + h0 + HK * BS, rDs + c * xlen + nc + HK * BS, nh - HK * BS, tid);
        }
    }
    __syncthreads();
    double dot[NRHS];
#pragma unroll
    for (int c = 0; c < NRHS; ++c) dot[c] = 0.0;
    for (int s = wave; s < nsl; s += NW) {
        uint32_t enext[PRE];
        int wnext = 0, e0next = 0;
        if (s + NW < nsl) fetch(s + NW, enext, e0next, wnext);
        else {
#pragma unroll
            for (int j = 0; j < PRE; ++j) enext[j] = padEnt;
        }
        const int i = s * 64 + lane;
        const bool live = i < nc;
        const int gi = c0 + (live ? i : 0);
        double xi[NV], acc[NV], accS[OP == OP_PROLOGUE ? NRHS : 1];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            xi[v] = (OP != OP_SUMA && live && act[v % NRHS]) ? xs[v * xlen + i] : 0.0;
            const double dgv = (OP == OP_AINV) ? 0.0 : (s == wave ? dg0[v % NRHS] : V.diag[v % NRHS][gi]);
            if (MUL) acc[v] = dgv * xi[v];
            else if (OP == OP_SUMA) acc[v] = dgv;
            else acc[v] = 0.0;
            if (OP == OP_PROLOGUE && v < NRHS) accS[v] = dgv;
        }
        auto accumulate = [&](uint32_t en) {
            const int o = en & 0xFFFFu, sl = (en >> 16) & 0x7FFFu;
            const bool lowerSide = (en >> 31) != 0u;
            const double cu = cU[sl], cl = cL[sl];
            const double cPlain = lowerSide ? cl : cu, cTrans = lowerSide ? cu : cl;   // TRANS swaps which triangle a side uses
            if (OP == OP_AINV) {
                if (sl < ifs0) { // faces only (AINVPreconditioner.C)
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const double rd = rDs[(SRD ? 0 : (v % NRHS)) * xlen + o];
                        acc[v] = fma((v < NRHS ? cPlain : cTrans) * rd, xs[v * xlen + o], acc[v]);
                    }
                }
            } else if (OP == OP_SUMA) {
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += cPlain;
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] = fma(v < NRHS ? cPlain : cTrans, xs[v * xlen + o], acc[v]);
                if (OP == OP_PROLOGUE) {
#pragma unroll
                    for (int c = 0; c < NRHS; ++c) accS[c] += cPlain;
                }
            }
        };
#pragma unroll
        for (int j = 0; j < PRE; ++j) if (j < wcur) accumulate(ecur[j]);
        if (wcur > PRE) {
            const uint32_t* ent = a.entries + e0cur + lane;
            for (int j = PRE; j < wcur; ++j) accumulate(ent[j * 64]);
        }
        if (live) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (!act[v % NRHS]) continue;
                const double out = (OP == OP_AINV) ? rDs[(SRD ? 0 : (v % NRHS)) * xlen + i] * (xi[v] - acc[v]) : acc[v];
                if (V.y[v]) V.y[v][gi] = out;
                if (MUL && V.y2[v]) V.y2[v][gi] = V.b[v % NRHS][gi] - out;
                if (OP == OP_PROLOGUE && v < NRHS && V.sumA[v]) V.sumA[v][gi] = accS[v];
                if (OP != OP_SUMA && v < NRHS) dot[v] = fma(out, xi[(NRHS + v) % NV], dot[v]);
            }
        }
#pragma unroll
        for (int j = 0; j < PRE; ++j) ecur[j] = enext[j];
        wcur = wnext; e0cur = e0next;
    }
    if (OP != OP_SUMA && V.dotPartial[0]) {
        __shared__ double red[BS / 64];
#pragma unroll
        for (int c = 0; c < NRHS; ++c) {
            const double tsum = block_sum<BS>(dot[c], red);
            if (tid == 0 && act[c]) V.dotPartial[c][b] = tsum;
        }
    }
}

// fold n per-workgroup partials into the RG slots the consumers reduce (fixed order)
__global__ __launch_bounds__(1024) void k_fold_partials(const double* __restrict__ in, int n, double* __restrict__ out)
{
    double v = 0.0;
    for (int k = threadIdx.x; k < n; k += 1024) v += in[k];
    out[threadIdx.x] = v;
}

// the same for up to three arrays in one launch (block c: in[c] -> out[c])
struct Fold3 { const double* in[3]; double* out[3]; };
__global__ __launch_bounds__(1024) void k_fold_partials3(const Fold3 f, int n)
{
    const double* in = f.in[blockIdx.x];
    double v = 0.0;
    for (int k = threadIdx.x; k < n; k += 1024) v += in[k];
    f.out[blockIdx.x][threadIdx.x] = v;
}
// the same for two arrays in one launch (block 0: inA -> outA, block 1: inB -> outB)
__global__ __launch_bounds__(1024) void k_fold_partials2(const double* __restrict__ inA, const double* __restrict__ inB, int n,
                                                         double* __restrict__ outA, double* __restrict__ outB)
{
    const double* in = blockIdx.x == 0 ? inA : inB;
    double v = 0.0;
    for (int k = threadIdx.x; k < n; k += 1024) v += in[k];
    (blockIdx.x == 0 ? outA : outB)[threadIdx.x] = v;
}

// ---------------------------------------------------------------------------
// layout kernels
// ---------------------------------------------------------------------------
// perm == nullptr: engine order == caller order (ordered addressing) -- a plain 16-byte copy
__device__ __forceinline__ void copy_chunked(const double* __restrict__ in, double* __restrict__ out, int n)
{
    const bool al = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    const int stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (al) {
        const int n2 = n >> 1;
        for (int i = t; i < n2; i += stride) reinterpret_cast<double2*>(out)[i] = reinterpret_cast<const double2*>(in)[i];
        if ((n & 1) && t == 0) out[n - 1] = in[n - 1];
    } else for (int i = t; i < n; i += stride) out[i] = in[i];
}
__global__ void k_gather_perm(const double* __restrict__ in, const int32_t* __restrict__ perm, double* __restrict__ out, int n)
{
    if (!perm) { copy_chunked(in, out, n); return; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[perm[i]];
}
__global__ void k_scatter_perm(const double* __restrict__ in, const int32_t* __restrict__ perm, double* __restrict__ out, int n)
{
    if (!perm) { copy_chunked(in, out, n); return; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[perm[i]] = in[i];
}
// caller coefficients -> engine slots (K22 calcSortCoeffs, lduMatrix.C:388-401, generalised)
__global__ void k_fill_slots(const double* __restrict__ upper, const double* __restrict__ lower,
                             const int32_t* __restrict__ slotFace, double* __restrict__ upE,
                             double* __restrict__ lowE, int64_t nSlots)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nSlots; k += (int64_t)gridDim.x * blockDim.x) {
        const int f = slotFace[k];
        if (f >= 0) { upE[k] = upper[f]; if (lowE) lowE[k] = lower[f]; }
        else if (f == -1) { upE[k] = 0.0; if (lowE) lowE[k] = 0.0; }
        // f <= -2: interface slot, owned by k_fill_iface
    }
}
__global__ void k_fill_iface(const double* __restrict__ bou, const double* __restrict__ inte,
                             const int32_t* __restrict__ extSlot, double* __restrict__ upE,
                             double* __restrict__ lowE, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int k = extSlot[i];
        upE[k] = -bou[i];
        if (lowE) lowE[k] = inte ? -inte[i] : -bou[i];
    }
}
__global__ void k_halo_pack(const double* __restrict__ x, const int32_t* __restrict__ cells, double* __restrict__ send, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) send[i] = x[cells[i]];
}
// patchNeighbourField: local (cyclic) faces read the partner cell, remote faces the received ext value
__global__ void k_patch_nbr_field(const double* __restrict__ psi, const int32_t* __restrict__ nbrCell, const double* __restrict__ ext,
                                  double* __restrict__ out, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = nbrCell[i];
        out[i] = c >= 0 ? psi[c] : ext[i];
    }
}
__global__ void k_faceH(const double* __restrict__ psi, const int32_t* __restrict__ lo, const int32_t* __restrict__ up,
                        const int32_t* __restrict__ faceSlot, const double* __restrict__ upE,
                        const double* __restrict__ lowE, double* __restrict__ out, int nFaces)
{
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nFaces; f += gridDim.x * blockDim.x) {
        const int k = faceSlot[f];
        out[f] = fma(upE[k], psi[up[f]], -(lowE[k] * psi[lo[f]]));   // lduMatrixfaceHFunctor (lduMatrixTemplates.C:38-48) as the compiled reference rounds it
    }
}

// ---------------------------------------------------------------------------
// streaming vector kernels.  Fixed grid RG x RB; block b owns the contiguous
// chunk [b*chunk, (b+1)*chunk) (chunk even), threads walk it in double2.
// ---------------------------------------------------------------------------
struct Chunk { int64_t lo, hi; };
__device__ __forceinline__ Chunk my_chunk(int64_t n)
{
    int64_t chunk = (n + RG - 1) / RG;
    chunk = (chunk + 1) & ~int64_t(1);
    Chunk c;
    c.lo = (int64_t)blockIdx.x * chunk; if (c.lo > n) c.lo = n;
    c.hi = c.lo + chunk; if (c.hi > n) c.hi = n;
    return c;
}
// iterate pairs then the odd tail; body2 sees an even-aligned index i (use 16-byte
// loads at i), body1 the single trailing element of an odd-length chunk.
template <class F2, class F1>
__device__ __forceinline__ void chunk_loop(int64_t n, F2 body2, F1 body1)
{
    const Chunk ck = my_chunk(n);
    const int64_t np = (ck.hi - ck.lo) >> 1;
#pragma unroll 4
    for (int64_t q = threadIdx.x; q < np; q += RB) body2(ck.lo + 2 * q);
    if (((ck.hi - ck.lo) & 1) && threadIdx.x == 0) body1(ck.hi - 1);
}
// the same, unrolled twice only: passes over many streams (k_pcg_update_psi_r: four reads, one write, two sums) fit 64 VGPRs with it
template <class F2, class F1>
__device__ __forceinline__ void chunk_loop2(int64_t n, F2 body2, F1 body1)
{
    const Chunk ck = my_chunk(n);
    const int64_t np = (ck.hi - ck.lo) >> 1;
#pragma unroll 2
    for (int64_t q = threadIdx.x; q < np; q += RB) body2(ck.lo + 2 * q);
    if (((ck.hi - ck.lo) & 1) && threadIdx.x == 0) body1(ck.hi - 1);
}
typedef double mi_dvec2 __attribute__((ext_vector_type(2)));
// 16-byte accesses of the streaming kernels.  Ordinary cache policy: non-temporal accesses on EVERY vector were measured
// (no gain, profiles/r02_b_cache_policy_ab.md); the two streams that are dead after one touch have their own helpers below.
__device__ __forceinline__ double2 ld2(const double* p, int64_t i) { return *reinterpret_cast<const double2*>(p + i); }
// rD (reciprocal diagonal) is read twice per PCG iteration and never written inside a solve
__device__ __forceinline__ double2 ld2_rd(const double* p, int64_t i) { return ld2(p, i); }
__device__ __forceinline__ void st2(double* p, int64_t i, double2 v) { *reinterpret_cast<double2*>(p + i) = v; }

// psi is touched once per Krylov iteration (read-modify-write) and the Amul result wA is dead once the residual update has
// read it: both are STREAMED (non-temporal), so that psi's dirty lines leave with the pass that wrote them instead of being
// written back under the Amul that follows (Amul inside the PCG loop 159 -> 151 us, 3 265 -> 3 365 it/s on one box).
__device__ __forceinline__ double2 ld2_stream(const double* p, int64_t i)
{
    const mi_dvec2 v = __builtin_nontemporal_load(reinterpret_cast<const mi_dvec2*>(p + i));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void st2_stream(double* p, int64_t i, double2 v)
{
    mi_dvec2 w; w.x = v.x; w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<mi_dvec2*>(p + i));
}

enum { RED_SUM = 0, RED_PROD = 1, RED_MAG = 2 };
template <int KIND>
__global__ __launch_bounds__(RB) void k_reduce(const double* __restrict__ a, const double* __restrict__ b, int64_t n, double* __restrict__ partial)
{
    __shared__ double red[RB / 64];
    double acc0 = 0, acc1 = 0;
    chunk_loop(n, [&](int64_t i) { const double2 x = ld2(a, i);
          if (KIND == RED_SUM) { acc0 += x.x; acc1 += x.y; }
          else if (KIND == RED_MAG) { acc0 += fabs(x.x); acc1 += fabs(x.y); }
          else { const double2 y = ld2(b, i); acc0 = fma(x.x, y.x, acc0); acc1 = fma(x.y, y.y, acc1); } },
        [&](int64_t i) { const double x = a[i];
          if (KIND == RED_SUM) acc0 += x; else if (KIND == RED_MAG) acc0 += fabs(x); else acc0 = fma(x, b[i], acc0); });
    const double t = block_sum<RB>(acc0 + acc1, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
// k_fold_partials followed by k_reduce_final in one launch, same summation order (distributed path:
// the local sum wA.pA goes straight into the all-reduce buffer)
__global__ __launch_bounds__(1024) void k_fold_final(const double* __restrict__ in, int n, double* __restrict__ out)
{
    __shared__ double folded[RG];
    __shared__ double red[RB / 64];
    double v = 0.0;
    for (int k = threadIdx.x; k < n; k += 1024) v += in[k];
    folded[threadIdx.x] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x < RB) {
#pragma unroll
        for (int k = 0; k < RG / RB; ++k) t += folded[threadIdx.x + k * RB];
    }
    t = wave_sum(t);
    const int wave = threadIdx.x >> 6;
    if (wave < RB / 64 && (threadIdx.x & 63) == 0) red[wave] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = red[0];
#pragma unroll
        for (int w = 1; w < RB / 64; ++w) r += red[w];
        *out = r;
    }
}
// two final reductions in one launch (block b reduces partial_b into out_b)
__global__ __launch_bounds__(RB) void k_reduce_final2(const double* __restrict__ partialA, double* __restrict__ outA,
                                                      const double* __restrict__ partialB, double* __restrict__ outB)
{
    __shared__ double red[RB / 64];
    const double t = sum_partials(blockIdx.x == 0 ? partialA : partialB, red);
    if (threadIdx.x == 0) *(blockIdx.x == 0 ? outA : outB) = t;
}
// communicator-attached solver loops: the all-reduced sum goes back into the partial array as {sum, 0, 0, ...}, so that every
// consumer's own re-reduction (sum_partials: same order in every block) returns exactly the global value (block b: set b)
__global__ __launch_bounds__(RB) void k_spread_partials(const double* __restrict__ sums, double* __restrict__ PA, double* __restrict__ PB)
{
    double* P = blockIdx.x == 0 ? PA : PB;
    const double v = sums[blockIdx.x];
    for (int i = threadIdx.x; i < RG; i += RB) P[i] = (i == 0) ? v : 0.0;
}
__global__ __launch_bounds__(RB) void k_reduce_final(const double* __restrict__ partial, double* __restrict__ out)
{
    __shared__ double red[RB / 64];
    const double t = sum_partials(partial, red);
    if (threadIdx.x == 0) *out = t;
}

// out = a + s*b  (one explicit fma; s by value).  out may alias a or b element-wise (in-place updates), hence no __restrict__ on them
__global__ __launch_bounds__(RB) void k_xpsy(double* out, const double* a, double s, const double* b, int64_t n)
{
    chunk_loop(n, [&](int64_t i) { const double2 x = ld2(a, i), y = ld2(b, i); st2(out, i, make_double2(fma(s, y.x, x.x), fma(s, y.y, x.y))); },
        [&](int64_t i) { out[i] = fma(s, b[i], a[i]); });
}
// out = a - b
__global__ __launch_bounds__(RB) void k_sub(double* __restrict__ out, const double* __restrict__ a, const double* __restrict__ b, int64_t n)
{
    chunk_loop(n, [&](int64_t i) { const double2 x = ld2(a, i), y = ld2(b, i); st2(out, i, make_double2(x.x - y.x, x.y - y.y)); },
        [&](int64_t i) { out[i] = a[i] - b[i]; });
}
// out = a * b   (diagonal precondition, diagonalPreconditioner.C:74-89)
__global__ __launch_bounds__(RB) void k_mul(double* __restrict__ out, const double* __restrict__ a, const double* __restrict__ b, int64_t n)
{
    chunk_loop(n, [&](int64_t i) { const double2 x = ld2(a, i), y = ld2(b, i); st2(out, i, make_double2(x.x * y.x, x.y * y.y)); },
        [&](int64_t i) { out[i] = a[i] * b[i]; });
}
// out = 1/a   (rD, diagonalPreconditioner.C:61-67)
__global__ __launch_bounds__(RB) void k_recip(double* __restrict__ out, const double* __restrict__ a, int64_t n)
{
    chunk_loop(n, [&](int64_t i) { const double2 x = ld2(a, i); st2(out, i, make_double2(1.0 / x.x, 1.0 / x.y)); },
        [&](int64_t i) { out[i] = 1.0 / a[i]; });
}
// normFactor partials (lduMatrixSolver.C:183-202): |Apsi - avg*sumA| + |b - avg*sumA|
__global__ __launch_bounds__(RB) void k_normfactor(const double* __restrict__ Apsi, const double* __restrict__ src,
                                                   const double* __restrict__ sumA, double avg, int64_t n, double* __restrict__ partial)
{
    __shared__ double red[RB / 64];
    double acc0 = 0, acc1 = 0;
    chunk_loop(n, [&](int64_t i) { const double2 A = ld2(Apsi, i), b = ld2(src, i), s = ld2(sumA, i);
          const double t0 = avg * s.x, t1 = avg * s.y;
          acc0 += fabs(A.x - t0) + fabs(b.x - t0); acc1 += fabs(A.y - t1) + fabs(b.y - t1); },
        [&](int64_t i) { const double t0 = avg * sumA[i]; acc0 += fabs(Apsi[i] - t0) + fabs(src[i] - t0); });
    const double t = block_sum<RB>(acc0 + acc1, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// the same with gAverage(psi) = sum(psi) / n formed on the device from the reduced sum (no host read in the solver prologue)
__global__ __launch_bounds__(RB) void k_normfactor_dev(const double* __restrict__ Apsi, const double* __restrict__ src,
                                                       const double* __restrict__ sumA, const double* __restrict__ sumPsi, double nGlobal,
                                                       int64_t n, double* __restrict__ partial)
{
    __shared__ double red[RB / 64];
    const double avg = sumPsi[0] / nGlobal;
    double acc0 = 0, acc1 = 0;
    chunk_loop(n, [&](int64_t i) { const double2 A = ld2(Apsi, i), b = ld2(src, i), s = ld2(sumA, i);
          const double t0 = avg * s.x, t1 = avg * s.y;
          acc0 += fabs(A.x - t0) + fabs(b.x - t0); acc1 += fabs(A.y - t1) + fabs(b.y - t1); },
        [&](int64_t i) { const double t0 = avg * sumA[i]; acc0 += fabs(Apsi[i] - t0) + fabs(src[i] - t0); });
    const double t = block_sum<RB>(acc0 + acc1, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// normFactor partials AND the partials of gSumMag(source - Apsi) in one pass over Apsi / source / sumA: the residual is formed again
// from its two operands (the rounding of k_sub / of the prologue tile pass), each sum keeps the per-thread order of its own kernel
// (k_normfactor_dev, k_reduce<RED_MAG>): same partials bit for bit, one vector pass instead of two.  gAverage(psi) from the RG partials
// of its sum (SCALAR = false: every block re-reduces them in sum_partials' order, no finalize launch) or from the all-reduced scalar.
template <bool SCALAR>
__global__ __launch_bounds__(RB, 8) void k_normfactor_mag(const double* __restrict__ Apsi, const double* __restrict__ src, const double* __restrict__ sumA,
                                                       const double* __restrict__ sumPsi, double nGlobal, int64_t n,
                                                       double* __restrict__ partialNF, double* __restrict__ partialMag)
{
    __shared__ double red[RB / 64];
    const double avg = (SCALAR ? sumPsi[0] : sum_partials(sumPsi, red)) / nGlobal;
    double acc0 = 0, acc1 = 0, m0 = 0, m1 = 0;
    chunk_loop(n, [&](int64_t i) { const double2 A = ld2(Apsi, i), b = ld2(src, i), s = ld2(sumA, i);
          const double t0 = avg * s.x, t1 = avg * s.y;
          acc0 += fabs(A.x - t0) + fabs(b.x - t0); acc1 += fabs(A.y - t1) + fabs(b.y - t1);
          m0 += fabs(b.x - A.x); m1 += fabs(b.y - A.y); },
        [&](int64_t i) { const double t0 = avg * sumA[i]; acc0 += fabs(Apsi[i] - t0) + fabs(src[i] - t0); m0 += fabs(src[i] - Apsi[i]); });
    const double t = block_sum<RB>(acc0 + acc1, red);
    const double u = block_sum<RB>(m0 + m1, red);
    if (threadIdx.x == 0) { partialNF[blockIdx.x] = t; partialMag[blockIdx.x] = u; }
}
// the multi-vector solver's prologue: the same for up to three components in one launch (blockIdx.y = component), and the sums of
// their psi likewise
struct NormMag3 { const double *Apsi[3], *src[3], *sumA[3], *sumPsi[3]; double *partialNF[3], *partialMag[3]; };
template <bool SCALAR>
__global__ __launch_bounds__(RB, 8) void k_normfactor_mag3(const NormMag3 A3, double nGlobal, int64_t n)
{
    __shared__ double red[RB / 64];
    const int k = blockIdx.y;
    const double* __restrict__ Apsi = A3.Apsi[k]; const double* __restrict__ src = A3.src[k]; const double* __restrict__ sumA = A3.sumA[k];
    const double avg = (SCALAR ? A3.sumPsi[k][0] : sum_partials(A3.sumPsi[k], red)) / nGlobal;
    double acc0 = 0, acc1 = 0, m0 = 0, m1 = 0;
    chunk_loop(n, [&](int64_t i) { const double2 A = ld2(Apsi, i), b = ld2(src, i), s = ld2(sumA, i);
          const double t0 = avg * s.x, t1 = avg * s.y;
          acc0 += fabs(A.x - t0) + fabs(b.x - t0); acc1 += fabs(A.y - t1) + fabs(b.y - t1);
          m0 += fabs(b.x - A.x); m1 += fabs(b.y - A.y); },
        [&](int64_t i) { const double t0 = avg * sumA[i]; acc0 += fabs(Apsi[i] - t0) + fabs(src[i] - t0); m0 += fabs(src[i] - Apsi[i]); });
    const double t = block_sum<RB>(acc0 + acc1, red);
    const double u = block_sum<RB>(m0 + m1, red);
    if (threadIdx.x == 0) { A3.partialNF[k][blockIdx.x] = t; A3.partialMag[k][blockIdx.x] = u; }
}
struct Sum3 { const double* a[3]; double* partial[3]; };
__global__ __launch_bounds__(RB) void k_sum3(const Sum3 S, int64_t n)
{
    __shared__ double red[RB / 64];
    const double* __restrict__ a = S.a[blockIdx.y];
    double acc0 = 0, acc1 = 0;
    chunk_loop(n, [&](int64_t i) { const double2 x = ld2(a, i); acc0 += x.x; acc1 += x.y; }, [&](int64_t i) { acc0 += a[i]; });
    const double t = block_sum<RB>(acc0 + acc1, red);
    if (threadIdx.x == 0) S.partial[blockIdx.y][blockIdx.x] = t;
}

// ---------------------------------------------------------------------------
// device-resident PCG pipeline (PCG.C:133-204).  The host only enqueues; all
// scalars (wArA, beta, wApA, alpha, residual, loop condition) live in PcgState.
// Every kernel starts with `if (st->done) return;` so iterations enqueued past
// convergence are no-ops and the result is exactly the reference's loop.
// ---------------------------------------------------------------------------
struct PcgState {
    double wArA[2];  // indexed by iteration parity
    double alpha, wApA;
    double normFactor, initialResidual, finalResidual;
    double tolerance, relTol;
    int32_t maxIter, minIter;
    int32_t nIterations, done, converged, singular;
    int32_t it;        // device-side iteration counter: kernels launched with it < 0 (graph replays) read it here
    // deferred psi update (PCG): k_pcg_update_psi_r(it) leaves `psi += alpha pA` to the next k_pcg_update_p, which reads pA
    // anyway (one vector read less per iteration, same values).  rItP1 = it + 1 once the residual update of iteration it has
    // run (k_pcg_update_psi_r, block 0); pApplyItP1 = it once the psi term of iteration it - 1 has been added -- recorded by
    // the single-workgroup kernel that FOLLOWS the adding pass (k_pcg_final of the same iteration, k_pcg_flush_mark at the end
    // of a solve), never by the adding pass itself, whose blocks read it.
    int32_t rItP1, pApplyItP1;
    int32_t fault;     // a grid barrier of the persistent kernel (persist.inc) ran out of polls: the results are not valid
};

constexpr double SP_SMALL = 1e-20, SP_VSMALL = 1e-300, SP_GREAT = 1e20; // SolverPerformance.H:269-275
static_assert(FX_VSMALL == SP_VSMALL, "tile_body's scaling factor uses GAMGSolver::scale's stabilise(den, VSMALL)");

__device__ __forceinline__ bool sp_converged(const PcgState* st, double res)
{
    // SolverPerformance.C:60-92
    return (res < st->tolerance) || (st->relTol > SP_SMALL && res < st->relTol * st->initialResidual);
}

// wA = rD*rA (or rA), partial1 = sum wA*rA     [precondition + gSumProd, PCG.C:139-142]
template <bool HAVE_RD>
__global__ __launch_bounds__(RB) void k_pcg_precond_dot(const PcgState* __restrict__ st, const double* __restrict__ rD,
                                                        const double* __restrict__ rA, double* __restrict__ wA,
                                                        int64_t n, double* __restrict__ partial)
{
    if (st->done) return;
    __shared__ double red[RB / 64];
    double acc0 = 0, acc1 = 0;
    chunk_loop(n, [&](int64_t i) { const double2 r = ld2(rA, i); double2 w = r;
          if (HAVE_RD) { const double2 d = ld2(rD, i); w.x = d.x * r.x; w.y = d.y * r.y; }
          st2(wA, i, w); acc0 = fma(w.x, r.x, acc0); acc1 = fma(w.y, r.y, acc1); },
        [&](int64_t i) { const double r = rA[i]; const double w = HAVE_RD ? rD[i] * r : r; wA[i] = w; acc0 = fma(w, r, acc0); });
    const double t = block_sum<RB>(acc0 + acc1, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// wArA = sum(partial1); beta = wArA/wArAold; pA = wA (+ beta*pA)      [PCG.C:144-160]
// PMODE 0: wA is stored (any preconditioner); 1: wA = rD*rA recomputed on the fly (diagonal);
// 2: wA = rA (none) -- the precondition pass and the wA round trip through HBM are fused away.
// Fused into the head of this kernel (single-GPU path, partial3 != nullptr): the convergence test of
// iteration it-1 (k_pcg_final) -- every block evaluates it redundantly from the same partials, block 0
// records it; a separate launch per iteration is saved.
template <bool DIST>
__device__ __forceinline__ bool pcg_test_previous(PcgState* __restrict__ st, int itPrev, const double* __restrict__ partial3,
                                                  double* __restrict__ hist, int histLen, double* red)
{
    const bool sing = partial3[0] < 0.0; // sum|r| partials are never negative
    const double s = DIST ? partial3[0] : sum_partials(partial3, red); // DIST: global sum from the allreduce
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    if (sing) { if (lead) { st->singular = 1; st->done = 1; } return false; } // `break`: nIterations not incremented
    const double res = s / st->normFactor;
    const bool conv = sp_converged(st, res);
    const bool cont = (itPrev < st->maxIter && !conv) || (itPrev + 1 < st->minIter);
    if (lead) {
        st->finalResidual = res;
        if (itPrev + 1 < histLen) hist[itPrev + 1] = res;
        st->nIterations = itPrev + 1;
        st->converged = conv;
        if (!cont) st->done = 1;
    }
    return cont;
}

// the pass as a device function (returns false when the launch had nothing to update: solve finished): the single-GPU kernel
// below and the fused distributed kernel (peer.inc: + halo pack into the neighbours' windows) are both this body
template <int PMODE, bool DIST>
__device__ __forceinline__ bool pcg_update_p_body(PcgState* __restrict__ st, int it, const double* __restrict__ partial1,
                                                  const double* __restrict__ wA, const double* __restrict__ rD,
                                                  const double* __restrict__ rA, double* __restrict__ pA, int64_t n,
                                                  const double* __restrict__ partial3, double* __restrict__ hist,
                                                  int histLen, double* __restrict__ psi, double* red) // psi != nullptr: deferred psi update
{
    const int itk = it < 0 ? st->it : it; // graph replay: the counter lives on the device (advanced by k_pcg_final)
    // the psi term of iteration itk - 1 is still owed when its residual update ran (and, the solve having ended there or not,
    // must be added exactly once: here, or by k_pcg_flush_psi when no further k_pcg_update_p follows)
    // (pApplyItP1 is written by k_pcg_final / k_pcg_flush_mark only -- kernels that run strictly between two of these -- so
    //  every block of this launch reads the same value whenever it starts)
    const bool owed = psi != nullptr && itk > 0 && st->rItP1 == itk && st->pApplyItP1 != itk;
    auto add_owed = [&]() {
        const double alpha = st->alpha;
        chunk_loop(n, [&](int64_t i) { const double2 p = ld2(pA, i); double2 x = ld2_stream(psi, i); x.x = fma(alpha, p.x, x.x); x.y = fma(alpha, p.y, x.y); st2_stream(psi, i, x); },
                   [&](int64_t i) { psi[i] = fma(alpha, pA[i], psi[i]); });
    };
    if (st->done) { if (owed) add_owed(); return false; }
    it = itk;
    if (partial3 && it > 0 && !pcg_test_previous<DIST>(st, it - 1, partial3, hist, histLen, red)) { if (owed) add_owed(); return false; }
    const double wArA = DIST ? partial1[0] : sum_partials(partial1, red); // DIST: global sum from the allreduce
    const double beta = (it == 0) ? 0.0 : wArA / st->wArA[(it & 1) ^ 1];
    const bool first = (it == 0);
    const double alpha = owed ? st->alpha : 0.0;
    chunk_loop(n, [&](int64_t i) {
            double2 w;
            if (PMODE == 0) w = ld2(wA, i);
            else if (PMODE == 1) { const double2 d = ld2_rd(rD, i), r = ld2(rA, i); w = make_double2(d.x * r.x, d.y * r.y); }
            else w = ld2(rA, i);
            if (first) st2(pA, i, w);
            else {
                const double2 p = ld2(pA, i);
                if (owed) { double2 x = ld2_stream(psi, i); x.x = fma(alpha, p.x, x.x); x.y = fma(alpha, p.y, x.y); st2_stream(psi, i, x); }
                st2(pA, i, make_double2(fma(beta, p.x, w.x), fma(beta, p.y, w.y)));
            }
        },
        [&](int64_t i) {
            const double w = (PMODE == 0) ? wA[i] : (PMODE == 1) ? rD[i] * rA[i] : rA[i];
            if (owed) psi[i] = fma(alpha, pA[i], psi[i]);
            pA[i] = first ? w : fma(beta, pA[i], w);
        });
    if (blockIdx.x == 0 && threadIdx.x == 0) st->wArA[it & 1] = wArA;
    return true;
}
template <int PMODE, bool DIST = false>
__global__ __launch_bounds__(RB) void k_pcg_update_p(PcgState* __restrict__ st, int it, const double* __restrict__ partial1,
                                                     const double* __restrict__ wA, const double* __restrict__ rD,
                                                     const double* __restrict__ rA, double* __restrict__ pA, int64_t n,
                                                     const double* __restrict__ partial3 = nullptr, double* __restrict__ hist = nullptr,
                                                     int histLen = 0, double* __restrict__ psi = nullptr) // psi != nullptr: deferred psi update
{
    __shared__ double red[RB / 64];
    (void)pcg_update_p_body<PMODE, DIST>(st, it, partial1, wA, rD, rA, pA, n, partial3, hist, histLen, psi, red);
}
// end of a solve: the psi term of the last iteration whose residual update ran, unless a k_pcg_update_p already added it
__global__ __launch_bounds__(RB) void k_pcg_flush_psi(const PcgState* __restrict__ st, const double* __restrict__ pA, double* __restrict__ psi, int64_t n)
{
    if (st->rItP1 <= 0 || st->pApplyItP1 == st->rItP1) return;
    const double alpha = st->alpha;
    chunk_loop(n, [&](int64_t i) { const double2 p = ld2(pA, i); double2 x = ld2(psi, i); x.x = fma(alpha, p.x, x.x); x.y = fma(alpha, p.y, x.y); st2(psi, i, x); },
               [&](int64_t i) { psi[i] = fma(alpha, pA[i], psi[i]); });
}
__global__ void k_pcg_flush_mark(PcgState* __restrict__ st) { st->pApplyItP1 = st->rItP1; }

// wApA = sum(partial2); singular? ; alpha; psi += alpha pA; rA -= alpha wA; partial3 = sum|rA|  [PCG.C:166-195]
// PMODE 1/2 additionally produce partial1 = sum (M^-1 rA)*rA for the NEXT iteration's wArA
// (precondition + gSumProd of PCG.C:139-142 fused into this pass).
// (device body: returns false when the solve has finished -- nothing was written; true also on the singular exit, whose
//  partial3 markers still have to reach k_pcg_final / the other ranks)
template <int PMODE, bool DIST>
__device__ __forceinline__ bool pcg_update_psi_r_body(PcgState* __restrict__ st, int it, const double* __restrict__ partial2,
                                                      const double* __restrict__ pA, const double* __restrict__ wA,
                                                      const double* __restrict__ rD,
                                                      double* __restrict__ psi, double* __restrict__ rA, int64_t n,
                                                      double* __restrict__ partial3, double* __restrict__ partial1, int deferPsi, double* red)
{
    if (st->done) return false;
    if (it < 0) it = st->it;
    const double wApA = DIST ? partial2[0] : sum_partials(partial2, red);
    if (fabs(wApA) / st->normFactor < SP_VSMALL) { // checkSingularity, SolverPerformance.C:32-44
        // every block takes this branch; only later kernels read done/singular
        if (threadIdx.x == 0) partial3[blockIdx.x] = -1.0; // marks "singular" for k_pcg_final
        return true;
    }
    const double alpha = st->wArA[it & 1] / wApA;
    double acc0 = 0, acc1 = 0, d0 = 0, d1 = 0;
#if defined(MI_PSIR_UNROLL2)
    chunk_loop2(n, [&](int64_t i) {
#else
    chunk_loop(n, [&](int64_t i) {
#endif
            const double2 w = ld2_stream(wA, i);
            double2 r = ld2(rA, i);
            if (!deferPsi) { const double2 p = ld2(pA, i); double2 x = ld2(psi, i); x.x = fma(alpha, p.x, x.x); x.y = fma(alpha, p.y, x.y); st2(psi, i, x); }
            r.x = fma(-alpha, w.x, r.x); r.y = fma(-alpha, w.y, r.y);
            st2(rA, i, r); acc0 += fabs(r.x); acc1 += fabs(r.y);
            if (PMODE == 1) { const double2 d = ld2_rd(rD, i); d0 = fma(d.x * r.x, r.x, d0); d1 = fma(d.y * r.y, r.y, d1); }
            else if (PMODE == 2) { d0 = fma(r.x, r.x, d0); d1 = fma(r.y, r.y, d1); }
        },
        [&](int64_t i) {
            if (!deferPsi) psi[i] = fma(alpha, pA[i], psi[i]);
            const double r = fma(-alpha, wA[i], rA[i]); rA[i] = r; acc0 += fabs(r);
            if (PMODE == 1) d0 = fma(rD[i] * r, r, d0); else if (PMODE == 2) d0 = fma(r, r, d0);
        });
    const double t = block_sum<RB>(acc0 + acc1, red);
    if (threadIdx.x == 0) partial3[blockIdx.x] = t;
    if (PMODE != 0) {
        const double u = block_sum<RB>(d0 + d1, red);
        if (threadIdx.x == 0) partial1[blockIdx.x] = u;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->alpha = alpha; st->wApA = wApA; if (deferPsi) st->rItP1 = it + 1; }
    return true;
}
template <int PMODE, bool DIST = false>
#ifndef MI_PSIR_WAVES
#define MI_PSIR_WAVES 1
#endif
__global__ __launch_bounds__(RB, MI_PSIR_WAVES) void k_pcg_update_psi_r(PcgState* __restrict__ st, int it, const double* __restrict__ partial2,
                                                         const double* __restrict__ pA, const double* __restrict__ wA,
                                                         const double* __restrict__ rD,
                                                         double* __restrict__ psi, double* __restrict__ rA, int64_t n,
                                                         double* __restrict__ partial3, double* __restrict__ partial1, int deferPsi = 0)
{
    __shared__ double red[RB / 64];
    (void)pcg_update_psi_r_body<PMODE, DIST>(st, it, partial2, pA, wA, rD, psi, rA, n, partial3, partial1, deferPsi, red);
}

// ---------------------------------------------------------------------------
// Device-resident PBiCG (PBiCG.C:67-246): same scheme as PCG -- scalars in PcgState (wArA[] holds wArT),
// per-block partials re-reduced by the consumers, kernels past convergence exit at once.
// ---------------------------------------------------------------------------
// wA = rD*rA, wT = rD*rT (or copies), partial1 = sum wA*rT          [PBiCG.C:149-155, diagonal / no preconditioner]
template <bool HAVE_RD>
__global__ __launch_bounds__(RB) void k_bicg_precond_dot(const PcgState* __restrict__ st, const double* __restrict__ rD,
                                                         const double* __restrict__ rA, const double* __restrict__ rT,
                                                         double* __restrict__ wA, double* __restrict__ wT, int64_t n,
                                                         double* __restrict__ partial)
{
    if (st->done) return;
    __shared__ double red[RB / 64];
    double acc0 = 0, acc1 = 0;
    chunk_loop(n, [&](int64_t i) {
            const double2 a = ld2(rA, i), t = ld2(rT, i); double2 w = a, v = t;
            if (HAVE_RD) { const double2 d = ld2(rD, i); w.x = d.x * a.x; w.y = d.y * a.y; v.x = d.x * t.x; v.y = d.y * t.y; }
            st2(wA, i, w); st2(wT, i, v); acc0 = fma(w.x, t.x, acc0); acc1 = fma(w.y, t.y, acc1); },
        [&](int64_t i) { const double a = rA[i], t = rT[i]; const double w = HAVE_RD ? rD[i] * a : a, v = HAVE_RD ? rD[i] * t : t;
            wA[i] = w; wT[i] = v; acc0 = fma(w, t, acc0); });
    const double s = block_sum<RB>(acc0 + acc1, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// wArT = sum(partial1); beta; pA = wA + beta pA; pT = wT + beta pT       [PBiCG.C:157-175]
__global__ __launch_bounds__(RB) void k_bicg_update_p(PcgState* __restrict__ st, int it, const double* __restrict__ partial1,
                                                      const double* __restrict__ wA, const double* __restrict__ wT,
                                                      double* __restrict__ pA, double* __restrict__ pT, int64_t n)
{
    if (st->done) return;
    __shared__ double red[RB / 64];
    const double wArT = sum_partials(partial1, red);
    const bool first = (it == 0);
    const double beta = first ? 0.0 : wArT / st->wArA[(it & 1) ^ 1];
    chunk_loop(n, [&](int64_t i) {
            const double2 w = ld2(wA, i), v = ld2(wT, i);
            if (first) { st2(pA, i, w); st2(pT, i, v); }
            else { const double2 p = ld2(pA, i), q = ld2(pT, i);
                   st2(pA, i, make_double2(fma(beta, p.x, w.x), fma(beta, p.y, w.y))); st2(pT, i, make_double2(fma(beta, q.x, v.x), fma(beta, q.y, v.y))); } },
        [&](int64_t i) { pA[i] = first ? wA[i] : fma(beta, pA[i], wA[i]); pT[i] = first ? wT[i] : fma(beta, pT[i], wT[i]); });
    if (blockIdx.x == 0 && threadIdx.x == 0) st->wArA[it & 1] = wArT;
}
// wApT = sum(partial2); singular?; alpha; psi += alpha pA; rA -= alpha wA; rT -= alpha wT; partial3 = sum|rA|  [PBiCG.C:177-215]
__global__ __launch_bounds__(RB) void k_bicg_update_psi_r(PcgState* __restrict__ st, int it, const double* __restrict__ partial2,
                                                          const double* __restrict__ pA, const double* __restrict__ wA,
                                                          const double* __restrict__ wT, double* __restrict__ psi,
                                                          double* __restrict__ rA, double* __restrict__ rT, int64_t n,
                                                          double* __restrict__ partial3)
{
    if (st->done) return;
    __shared__ double red[RB / 64];
    const double wApT = sum_partials(partial2, red);
    if (fabs(wApT) / st->normFactor < SP_VSMALL) {
        if (threadIdx.x == 0) partial3[blockIdx.x] = -1.0; // "singular" marker for k_pcg_final
        return;
    }
    const double alpha = st->wArA[it & 1] / wApT;
    double acc0 = 0, acc1 = 0;
    chunk_loop(n, [&](int64_t i) {
            const double2 p = ld2(pA, i), w = ld2(wA, i), v = ld2(wT, i); double2 x = ld2_stream(psi, i), r = ld2(rA, i), t = ld2(rT, i);
            x.x = fma(alpha, p.x, x.x); x.y = fma(alpha, p.y, x.y);
            r.x = fma(-alpha, w.x, r.x); r.y = fma(-alpha, w.y, r.y);
            t.x = fma(-alpha, v.x, t.x); t.y = fma(-alpha, v.y, t.y);
            st2_stream(psi, i, x); st2(rA, i, r); st2(rT, i, t); acc0 += fabs(r.x); acc1 += fabs(r.y); },
        [&](int64_t i) { psi[i] = fma(alpha, pA[i], psi[i]); const double r = fma(-alpha, wA[i], rA[i]); rA[i] = r;
            rT[i] = fma(-alpha, wT[i], rT[i]); acc0 += fabs(r); });
    const double s = block_sum<RB>(acc0 + acc1, red);
    if (threadIdx.x == 0) partial3[blockIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->alpha = alpha; st->wApA = wApT; }
}

// ---------------------------------------------------------------------------
// Device-resident PBiCGStab (PBiCGStab.C:67-300).  PcgState: wArA[] = rA0rA (by iteration parity), alpha, wApA = omega.
// A kernel that decides something from data every workgroup recomputes the decision itself (same partials => same bits);
// `done` is only ever written by single-workgroup kernels or by workgroups that all take the same early exit.
// ---------------------------------------------------------------------------
// rA0rA = sum(P1); singular tests; beta = (rA0rA/rA0rAold)*(alpha/omega); pA = rA + beta*(pA - omega*AyA)   [:150-190]
__global__ __launch_bounds__(RB) void k_stab_update_p(PcgState* __restrict__ st, int it, const double* __restrict__ partial1,
                                                      const double* __restrict__ rA, const double* __restrict__ AyA,
                                                      double* __restrict__ pA, int64_t n)
{
    if (st->done) return;
    __shared__ double red[RB / 64];
    const double rA0rA = sum_partials(partial1, red);
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    if (fabs(rA0rA) < SP_VSMALL) { if (lead) { st->singular = 1; st->done = 1; } return; }       // checkSingularity, break
    if (it == 0) {
        chunk_loop(n, [&](int64_t i) { st2(pA, i, ld2(rA, i)); }, [&](int64_t i) { pA[i] = rA[i]; });
    } else {
        const double omega = st->wApA, alpha = st->alpha;
        if (fabs(omega) < SP_VSMALL) { if (lead) { st->singular = 1; st->done = 1; } return; }
        const double beta = (rA0rA / st->wArA[(it & 1) ^ 1]) * (alpha / omega);
        chunk_loop(n, [&](int64_t i) {
                const double2 r = ld2(rA, i), y = ld2(AyA, i), p = ld2(pA, i);
                st2(pA, i, make_double2(fma(beta, fma(-omega, y.x, p.x), r.x), fma(beta, fma(-omega, y.y, p.y), r.y))); },
            [&](int64_t i) { pA[i] = fma(beta, fma(-omega, AyA[i], pA[i]), rA[i]); });
    }
    if (lead) st->wArA[it & 1] = rA0rA;
}
// alpha = rA0rA / sum(P2); sA = rA - alpha*AyA; P3 = sum|sA|                                   [:200-215]
__global__ __launch_bounds__(RB) void k_stab_s(PcgState* __restrict__ st, int it, const double* __restrict__ partial2,
                                               const double* __restrict__ rA, const double* __restrict__ AyA,
                                               double* __restrict__ sA, int64_t n, double* __restrict__ partial3)
{
    if (st->done) return;
    __shared__ double red[RB / 64];
    const double alpha = st->wArA[it & 1] / sum_partials(partial2, red);
    double acc0 = 0, acc1 = 0;
    chunk_loop(n, [&](int64_t i) { const double2 r = ld2(rA, i), y = ld2(AyA, i);
            const double2 sv = make_double2(fma(-alpha, y.x, r.x), fma(-alpha, y.y, r.y)); st2(sA, i, sv); acc0 += fabs(sv.x); acc1 += fabs(sv.y); },
        [&](int64_t i) { const double sv = fma(-alpha, AyA[i], rA[i]); sA[i] = sv; acc0 += fabs(sv); });
    const double t = block_sum<RB>(acc0 + acc1, red);
    if (threadIdx.x == 0) partial3[blockIdx.x] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->alpha = alpha;
}
// mid-iteration exit: if sum|sA|/normFactor has converged, psi += alpha*yA                       [:217-232]
__global__ __launch_bounds__(RB) void k_stab_mid(const PcgState* __restrict__ st, const double* __restrict__ partial3,
                                                 const double* __restrict__ yA, double* __restrict__ psi, int64_t n)
{
    if (st->done) return;
    __shared__ double red[RB / 64];
    const double res = sum_partials(partial3, red) / st->normFactor;
    if (!sp_converged(st, res)) return;
    const double alpha = st->alpha;
    chunk_loop(n, [&](int64_t i) { const double2 y = ld2(yA, i); double2 x = ld2(psi, i); x.x = fma(alpha, y.x, x.x); x.y = fma(alpha, y.y, x.y); st2(psi, i, x); },
        [&](int64_t i) { psi[i] = fma(alpha, yA[i], psi[i]); });
}
__global__ __launch_bounds__(RB) void k_stab_mid_final(PcgState* __restrict__ st, int it, const double* __restrict__ partial3,
                                                       double* __restrict__ hist, int histLen)
{
    if (st->done) return;
    __shared__ double red[RB / 64];
    const double res = sum_partials(partial3, red) / st->normFactor;
    if (threadIdx.x != 0) return;
    st->finalResidual = res;
    if (sp_converged(st, res)) { // nIterations++; return (the residual of this half step is the last history entry)
        st->converged = 1; st->nIterations = it + 1; st->done = 1;
        if (it + 1 < histLen) hist[it + 1] = res;
    }
}
// two inner products of one vector in one pass: P4 = sum a*a, P5 = sum a*b                        [:240-247]
__global__ __launch_bounds__(RB) void k_reduce_two(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                                   double* __restrict__ partialAA, double* __restrict__ partialAB)
{
    __shared__ double red[RB / 64];
    double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    chunk_loop(n, [&](int64_t i) { const double2 x = ld2(a, i), y = ld2(b, i);
            a0 = fma(x.x, x.x, a0); a1 = fma(x.y, x.y, a1); b0 = fma(x.x, y.x, b0); b1 = fma(x.y, y.y, b1); },
        [&](int64_t i) { const double x = a[i]; a0 = fma(x, x, a0); b0 = fma(x, b[i], b0); });
    const double t = block_sum<RB>(a0 + a1, red);
    const double u = block_sum<RB>(b0 + b1, red);
    if (threadIdx.x == 0) { partialAA[blockIdx.x] = t; partialAB[blockIdx.x] = u; }
}
// omega = tAsA/tAtA; psi += alpha*yA; psi += omega*q; rA = sA - omega*tA; P3 = sum|rA|            [:249-285]
__global__ __launch_bounds__(RB) void k_stab_update(PcgState* __restrict__ st, const double* __restrict__ partialTT,
                                                    const double* __restrict__ partialTS, const double* __restrict__ yA,
                                                    const double* __restrict__ q, const double* __restrict__ sA,
                                                    const double* __restrict__ tA, double* __restrict__ psi,
                                                    double* __restrict__ rA, int64_t n, double* __restrict__ partial3)
{
    if (st->done) return;
    __shared__ double red[RB / 64];
    const double tAtA = sum_partials(partialTT, red);
    const double tAsA = sum_partials(partialTS, red);
    const double omega = tAsA / tAtA, alpha = st->alpha;
    double acc0 = 0, acc1 = 0;
    chunk_loop(n, [&](int64_t i) {
            const double2 y = ld2(yA, i), z = ld2(q, i), sv = ld2(sA, i), t = ld2(tA, i); double2 x = ld2_stream(psi, i);
            x.x = fma(omega, z.x, fma(alpha, y.x, x.x)); x.y = fma(omega, z.y, fma(alpha, y.y, x.y));
            const double2 r = make_double2(fma(-omega, t.x, sv.x), fma(-omega, t.y, sv.y));
            st2_stream(psi, i, x); st2(rA, i, r); acc0 += fabs(r.x); acc1 += fabs(r.y); },
        [&](int64_t i) { psi[i] = fma(omega, q[i], fma(alpha, yA[i], psi[i])); const double r = fma(-omega, tA[i], sA[i]); rA[i] = r; acc0 += fabs(r); });
    const double t = block_sum<RB>(acc0 + acc1, red);
    if (threadIdx.x == 0) partial3[blockIdx.x] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->wApA = omega;
}

// residual, history, do-while condition                                  [PCG.C:195-204]
template <bool DIST = false>
__global__ __launch_bounds__(RB) void k_pcg_final(PcgState* __restrict__ st, int it, const double* __restrict__ partial3,
                                                  double* __restrict__ hist, int histLen)
{
    if (it < 0) it = st->it;
    // the k_pcg_update_p of this iteration (it ran before this kernel, converged or not) has added the psi term iteration
    // it - 1 owed (rItP1 is it + 1 by now if this iteration's residual update ran, it if it was gated or singular)
    if (threadIdx.x == 0 && st->rItP1 >= it) st->pApplyItP1 = it;
    if (st->done) return;
    __shared__ double red[RB / 64];
    const bool sing = partial3[0] < 0.0; // sum|r| partials are never negative
    const double s = DIST ? partial3[0] : sum_partials(partial3, red);
    if (threadIdx.x != 0) return;
    if (sing) { st->singular = 1; st->done = 1; return; } // `break`: nIterations not incremented
    const double res = s / st->normFactor;
    st->finalResidual = res;
    if (it + 1 < histLen) hist[it + 1] = res;
    st->nIterations = it + 1;
    st->it = it + 1;
    const bool conv = sp_converged(st, res);
    st->converged = conv;
    const bool cont = (it < st->maxIter && !conv) || (it + 1 < st->minIter);
    if (!cont) st->done = 1;
}

// start of a solve: normFactor, initial residual, first convergence test   [PCG.C:105-121]
template <bool DIST>
__device__ __forceinline__ void solve_init_body(PcgState* __restrict__ st, const double* __restrict__ partialNF,
                                                const double* __restrict__ partialR, double* __restrict__ hist, int histLen, double* red)
{
    const double nf = (DIST ? partialNF[0] : sum_partials(partialNF, red)) + SP_SMALL;
    const double sr = DIST ? partialR[0] : sum_partials(partialR, red);
    if (threadIdx.x != 0) return;
    st->normFactor = nf;
    const double res = sr / nf;
    st->initialResidual = res; st->finalResidual = res;
    if (histLen > 0) hist[0] = res;
    st->nIterations = 0; st->singular = 0; st->it = 0;
    st->wArA[0] = SP_GREAT; st->wArA[1] = SP_GREAT;
    const bool conv = sp_converged(st, res);
    st->converged = conv;
    st->done = (st->minIter > 0 || !conv) ? 0 : 1;
}
template <bool DIST = false>
__global__ __launch_bounds__(RB) void k_solve_init(PcgState* __restrict__ st, const double* __restrict__ partialNF,
                                                   const double* __restrict__ partialR, double* __restrict__ hist, int histLen)
{
    __shared__ double red[RB / 64];
    solve_init_body<DIST>(st, partialNF, partialR, hist, histLen, red);
}
// up to three components' solves in one launch (block k: component k)
struct Init3 { PcgState* st[3]; const double *partialNF[3], *partialR[3]; double* hist[3]; };
template <bool DIST = false>
__global__ __launch_bounds__(RB) void k_solve_init3(const Init3 I, int histLen)
{
    __shared__ double red[RB / 64];
    const int k = blockIdx.x;
    solve_init_body<DIST>(I.st[k], I.partialNF[k], I.partialR[k], I.hist[k], histLen, red);
}

} // namespace mi
