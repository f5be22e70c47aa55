// engine.hip -- C-ABI (include/mi_ldu.h) of the MI355X-native lduMatrix engine.
// Host-side orchestration only; all arithmetic is in kernels.hip.hpp.
// There is no CPU fallback: without a gfx950 device every compute entry point
// returns MI_ERR_DEVICE.
#include <hip/hip_ext.h>
#include "../../include/mi_ldu.h"
#include "kernels.hip.hpp"
#include "tiling.hpp"
#include "host_parallel.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace mi;
constexpr int HALO_NV = 6;    // operand vectors one halo exchange carries (peer.inc: a window holds that many per parity; multi.inc: pA and pT of three components)

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(MI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));   \
    } while (0)
#define MICHK(expr) do { int r__ = (expr); if (r__ != MI_OK) return r__; } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    int alloc(size_t count)
    {
        if (count == n && p) return MI_OK;
        release();
        if (hipMalloc((void**)&p, (count ? count : 1) * sizeof(T)) != hipSuccess) { p = nullptr; return fail(MI_ERR_ALLOC, "hipMalloc failed"); }
        n = count;
        return MI_OK;
    }
    template <class A>
    int upload(const std::vector<T, A>& v, hipStream_t s)
    {
        MICHK(alloc(v.size()));
        if (!v.empty()) HIPCHK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
        return MI_OK;
    }
};

int env_int(const char* name, int def)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : def;
}

} // namespace

struct mi_ctx_s {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    DevBuf<double> partial;  // 4 * RG
    DevBuf<double> scalars;  // small device scalars
    DevBuf<PcgState> state;
    PcgState* hostState = nullptr; // pinned
    double* hostScal = nullptr;    // pinned
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t sideStream = nullptr;              // second stream (lazy): the GAMG coarsest-level inversion runs there beside the solve's prologue and first down-sweep
    hipEvent_t evSideGo = nullptr, evSideDone = nullptr;
    int amulBS = 0;
    int tileFlags = 0;
    int xcdRows = 1;      // MI_XCD_ROWS: XCD-aware block mapping of the caller-order row passes
    int persist = 0;      // MI_TILE_PERSIST: persistent tile launches (workgroups = resident slots, each walks a run of tiles)
    int nCU = 0;
    bool coarseLevelBuild = false; // set by the GAMG hierarchy builder around its level addressings (tile size choice)
    bool keepSlotTables = false;   // ... and: keep the host copies of slotFace / faceSlot (the builder derives its slot-to-slot children lists from them, then drops them)
    int attachEvents = 1; // MI_EVENT_ATTACH=0: plain hipEventRecord pairs around the Amul launch instead of kernel-attached events (A/B hook)
    int fusePerm = 1;  // MI_FUSE_PERM: caller-order operators gather / scatter through e2c inside the tile kernel (A/B hook)
    int deferPsi = 1;  // MI_PCG_DEFER_PSI: psi += alpha pA rides in the next k_pcg_update_p (one vector read less per iteration; A/B hook)
    int fuseFinal = 0; // MI_PCG_FUSE_FINAL: convergence test fused into the next update_p (A/B hook)
    int pcgFuseTest = 0; // tests ("pcg_fuse_test"): workgroups of the fused launch that leave its barrier at once (pcg_fused.inc)
    int pcgFuseRP = -1; // MI_PCG_FUSE_RP: residual update + next direction update as one launch (pcg_fused.inc); 0 never, -1 (default) once the device has been asked
    int pcgPersist = 1; // MI_PCG_PERSIST: 0 never, 1 (default) whenever the tiles fit the CUs' registers (persist.inc)
    int64_t stats[5] = {0, 0, 0, 0, 0}; // mi_ctx_stat
    int persistGrid = 0; // MI_PERSIST_GRID: workgroups of the persistent kernel (0: one per CU); MI_PERSIST_SHARED=1 lets ranks that share a device use it -- tests only: their grids must fit the device TOGETHER
    int persistShared = 0;
    int persistCoop = -1; // cooperative launch of the persistent kernel possible on this device AND its barrier litmus clean (-1: not asked yet)
    uint64_t faultEpoch = 0; // faults of the persistent kernel reported on this context so far (fetch_state); a matrix re-zeroes its barrier words when it has missed one
    int winDirect = 1; // MI_WIN_DIRECT: tile operators of attached matrices read neighbour-rank values straight from the halo window (one launch for all tiles) instead of k_halo_pull + a second launch (A/B hook)
    int gamgGraphAttached = 1; // MI_GAMG_GRAPH_ATTACHED: the V-cycle of a decomposed case replays as a hipGraph when every exchange of it is stream work (peer windows)
    int multiPipe = 1; // MI_MULTI_PIPE: tile_kernel_multi_pipe (multi_pipe.inc) for the multi-vector passes of the Krylov iterations
    int fusePrologue = 1; // MI_FUSE_PROLOGUE: A psi, source - A psi and sumA in one pass over the coefficients, the prologue's sums batched (A/B hook: 0 = the separate passes, same bits)
    int pairAT = 1;    // MI_PBICG_PAIR: PBiCG's A p / A^T pT (and the DILU pair) in one pass over the coefficients (A/B hook)
    struct mi_matrix_s* session = nullptr; // matrix whose mi_pcg_begin/iterate/end session owns this context's solver scratch (partial, scalars, state)
    int pcgBatch = 16, pcgGraph = -1, pbicgHostStepped = 0, gamgDeviceInvert = -1, gamgAlwaysAgglomerate = 0, gamgGraph = 1, gamgFuse = 1; // MI_* switches, read once per context
    std::set<const void*> ldsAttrSet;                         // kernels whose dynamic-LDS limit has been raised on THIS device
    std::map<std::pair<const void*, size_t>, int> occCache;  // (kernel, LDS bytes) -> resident workgroups per CU on this device
};

struct mi_addr_s {
    mi_ctx_s* ctx = nullptr;
    TileLayout L; // host copy (big vectors are dropped after upload except the permutations)
    DevBuf<int32_t> e2c, c2e, tileCellStart, tileSlotStart, tileIfaceSlot0, tileHaloStart, haloCell, tileSliceStart, sliceEntryStart;
    DevBuf<uint32_t> entries, entries16;
    DevBuf<int32_t> sliceEntryStart16, tileSbStart;
    DevBuf<uint16_t> slotBase;
    bool compact = false; // 16-bit row entries in use (MI_ENTRY16=1)
    DevBuf<int32_t> slotFace, extSlot, interiorTiles, boundaryTiles, patchFaceCellsE, faceSlot, lowerAddr, upperAddr;
    DevBuf<int32_t> ownerStartC, losortStartC, losortC; // caller-order row tables for the assembly sweeps (lazy)
    DevBuf<uint32_t> row16; DevBuf<uint16_t> losort16; DevBuf<int32_t> rowEsc, rowEscStart; // their block-local 16-bit form (assembly.inc: R16)
    struct RowPlan { bool tiles = false; int bs = 256, blocks = 0, maxFaces = 0, capForced = 0; } rowPlan[2]; // blocks of the assembly row passes: [0] sums / fused schemes, [1] gradient (assembly.inc)
    DevBuf<double> relaxD0, relaxSumOff;
    DevBuf<uint8_t> setMask; DevBuf<double> setVal;   // fvMatrix::setValues scratch (assembly.inc)
    Table<int32_t> lowerHost, upperHost; // kept for the lazily-built faceH tables
    int32_t nInterior = 0, nBoundary = 0, nLocalPatches = 0;
    Table<char> patchIsLocal; // [nPatches] cyclic (local) coupled patch: no exchange
    DevBuf<int32_t> ifaceNbrCaller; // [nExt] caller cell across every LOCAL interface face, -1 for remote faces (lazy)
    DevBuf<int32_t> haloSrc;        // [nHaloTot] caller cell of every halo entry, or -1-k for ext value k (lazy; caller-order tile launches)
    Table<Table<int32_t>> patchFaceCellsHost, patchNbrCellsHost; // caller order (GAMG interface agglomeration)
    int64_t nEntries = 0, nHaloTot = 0;
    // cyclicAMI patches (mi_addr_set_ami_patch): declared like processor patches (ext region), their neighbour values are
    // interpolated locally from the partner patch's cells before every operator that reads them
    struct AmiPatch {
        int32_t patch = 0, nbrPatch = 0, n = 0, extOff = 0;
        Table<int32_t> transports, partCount;   // partner SIDE split over several ranks: one transport patch per piece, faces per piece (transport == transports[0], nPartner == their sum)
        int32_t transport = -1, nPartner = 0;   // cyclicAMI whose partner patch lives on ANOTHER rank: the processor patch that carries the partner's internal field; partner patch size
        DevBuf<int32_t> start, cellE, ownE; DevBuf<double> w; bool hasLow = false;
        Table<int32_t> hStart, hAddr; Table<double> hW, hMagSf; // host copies: the GAMG builder agglomerates them
    };
    Table<AmiPatch*> ami;
    bool amiRemote = false;   // some cyclicAMI patch interpolates from a transport patch: interpolate AFTER the halo exchange
    ~mi_addr_s() { for (AmiPatch* q : ami) delete q; }
    bool identity = false; // engine order == caller order (ordered addressing, or a mesh whose numbering happens to be tile-contiguous)
    const int32_t* perm() const { return identity ? nullptr : e2c.p; } // nullptr: the permutation kernels degenerate to copies
};

struct mi_dpcg_s { double *psi = nullptr, *src = nullptr, *pA = nullptr, *wA = nullptr, *rA = nullptr, *scal = nullptr, *send = nullptr; int precond = 0; };

struct mi_matrix_s {
    mi_addr_s* addr = nullptr;
    mi_dpcg_s dp; // buffers of a distributed PCG session (owned by the caller)
    DevBuf<double> diagE, upE, lowE, rD, sumAE; // sumAE: lduMatrix::sumA of the bound coefficients (normFactor), kept per binding
    bool asym = false, bound = false, rDValid = false, sumAValid = false;
    Table<double> patchFactor; // transformCoupleField factor of every coupled patch (empty: none set, all 1)
    uint64_t epoch = 0; // bumped whenever coefficients are (re)bound: lets a GAMG hierarchy keep its level matrices between solves
    Table<DevBuf<double>*> work; // engine-order vectors (n_cells + n_ext)
    DevBuf<double> hist, tilePartial;
    DevBuf<PcgState> mstate; DevBuf<double> mpartial, mhist, mtilePartial; PcgState* mhostState = nullptr;   // multi-right-hand-side solves (multi.inc): one state / partial block / history per component
    DevBuf<double> persistScratch;   // per-workgroup partials + the grid barrier of the persistent PCG kernel (persist.inc)
    uint64_t persistFaultEpoch = 0;   // mi_ctx_s::faultEpoch when persistScratch was last zeroed
    DevBuf<double> persistZ;   // z = rD o rA published by the persistent PCG kernel (persist.inc, ZP)
    int histLen = 0;
    // running PCG session (mi_pcg_begin/iterate/end)
    int pcgIt = 0, pcgPrecond = MI_PRECOND_DIAGONAL;
    bool pcgActive = false;
    bool pcgPReady = false;            // pA of iteration pcgIt has been formed already (by the fused launch of the previous iteration, pcg_fused.inc)
    DevBuf<unsigned int> fusedBar;     // its barrier words, the generation of its last launch, the fault epoch the words were zeroed in
    unsigned int fusedGen = 0;
    Table<hipEvent_t> evPool;
    // hipGraph of one batch of device-resident PCG iterations (launch-bound regime: small meshes, coarse ranks)
    hipGraphExec_t pcgGraph = nullptr;
    struct { int precond = -1, batch = 0, histLen = 0; const void* hist = nullptr; const void* psi = nullptr; } pcgGraphKey;
    bool gateDone = false; // tile launches of a device-resident solver loop read PcgState::done and exit past convergence
    double *proY2 = nullptr, *proY3 = nullptr; // set around ONE OP_PROLOGUE tile pass (solve_prologue): rA = source - A psi and sumA go out of the same pass
    const double *callerX = nullptr, *callerB = nullptr; double* callerY = nullptr; // set around ONE tile launch: x, b, y are the caller's arrays (permutation folded into the kernel)
    hipEvent_t kevStart = nullptr, kevStop = nullptr; // when set: attached to the next tile-kernel launch (hipExtLaunchKernel)
    struct mi_dpcg_comm_s* dpc = nullptr; // attached RCCL communicators + exchange plan (comm.inc)
    DevBuf<double> sendBuf, dscal;         // halo send buffer / scalar block of engine-driven distributed solves
    ~mi_matrix_s();
    int vec(size_t k, double** out)
    {
        while (work.size() <= k) work.push_back(new DevBuf<double>());
        const size_t need = (size_t)addr->L.nCells + (size_t)addr->L.nExt;
        if (work[k]->n != need) {
            MICHK(work[k]->alloc(need));
            HIPCHK(hipMemsetAsync(work[k]->p, 0, need * sizeof(double), addr->ctx->stream));
        }
        *out = work[k]->p;
        return MI_OK;
    }
};

// communicator hooks (comm.inc): with a communicator attached to the matrix every operator that reads the
// coupled-patch neighbour values exchanges them itself and every global sum is all-reduced over the ranks
bool comm_remote(const mi_matrix_s* m);                         // attached and has processor patches
bool comm_attached(const mi_matrix_s* m);
bool comm_any_ami(const mi_matrix_s* m);                         // cyclicAMI patches here or -- agreed at attach time -- on any other rank of the case
bool comm_any_factor(const mi_matrix_s* m);                      // ... transformed patches (factor != 1) ...
bool comm_any_compact(const mi_matrix_s* m);                     // ... the 16-bit entry form ...
bool matrix_has_factor(const mi_matrix_s* m);
void comm_inherit_factor_flag(mi_matrix_s* level, const mi_matrix_s* fine);
int64_t comm_n_global(const mi_matrix_s* m);                     // global cell count (gAverage)
int comm_exchange_start(mi_matrix_s* m, const double* send, double* vec);
int comm_exchange_wait(mi_matrix_s* m);
int comm_allreduce(mi_matrix_s* m, double* dev, size_t n);
bool peer_halo_ready(const mi_matrix_s* m);                      // halo windows mapped by the neighbours (peer.inc)
int peer_exchange_push(mi_matrix_s* m, const double* x);         // x's patch values into the neighbours' windows
int peer_exchange_pull(mi_matrix_s* m, double* x);               // wait, then window -> x's ext region (patch factors applied)
int peer_exchange_push_n(mi_matrix_s* m, int nv, const double* const* xs);   // the same for up to HALO_NV operand vectors in ONE exchange (multi.inc)
int peer_exchange_pull_n(mi_matrix_s* m, int nv, double* const* xs);
bool peer_win_direct(mi_matrix_s* m);                            // boundary tiles may read the window themselves (no transformed patches, no AMI, default entries)
int peer_check(mi_matrix_s* m);                                  // MI_ERR_DEVICE when a window wait ran out of polls
int dpcg_flush_if_fused(mi_matrix_s* m);
bool peer_reduce_ready(const mi_matrix_s* m);                    // the matrix's all-reduces go through peer windows
int peer_globalize(mi_matrix_s* m, int n, double* const* P);     // n <= 6 arrays of RG block partials -> global sums, spread back as {sum, 0, ...}: ONE launch (peer.inc)
int pcg_solve_attached(mi_matrix_s* m, double* psi_io, const double* source, const mi_solver_controls* ctl, int precond,
                       mi_solver_perf* perf, double* hist_host, int32_t hist_len);

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
extern "C" const char* mi_last_error(void) { return g_err.c_str(); }

extern "C" int mi_device_available(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

extern "C" int mi_ctx_create(int device, void* hip_stream, mi_ctx_t* out)
{
    if (!out) return fail(MI_ERR_ARG, "out is NULL");
    int n = 0;
    const hipError_t e0 = hipGetDeviceCount(&n);
    if (e0 != hipSuccess || n <= 0)
        return fail(MI_ERR_DEVICE, std::string("no HIP device visible: the engine has no CPU fallback (hipGetDeviceCount: ") + hipGetErrorString(e0) + ", " + std::to_string(n) + " devices)");
    if (device < 0 || device >= n) return fail(MI_ERR_ARG, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MI_ERR_DEVICE, std::string("engine is built for gfx950 only, found ") + prop.gcnArchName);
    mi_ctx_s* c = new mi_ctx_s();
    c->device = device;
    if (hip_stream) c->stream = (hipStream_t)hip_stream;
    else { if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return fail(MI_ERR_DEVICE, "hipStreamCreate failed"); } c->ownStream = true; }
    int r = c->partial.alloc(4 * RG);
    if (r == MI_OK) r = c->scalars.alloc(16);
    if (r == MI_OK) r = c->state.alloc(1);
    if (r != MI_OK) { delete c; return r; }
    if (hipHostMalloc((void**)&c->hostState, sizeof(PcgState)) != hipSuccess ||
        hipHostMalloc((void**)&c->hostScal, 16 * sizeof(double)) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c; return fail(MI_ERR_DEVICE, "pinned host / event allocation failed");
    }
    c->tileFlags = env_int("MI_TILE_FLAGS", 1); // bit0: coefficient segments are staged with non-temporal loads (read once per launch): Amul -4 % (profiles/r02_b_cache_policy_ab.md)
    c->attachEvents = env_int("MI_EVENT_ATTACH", 1);
    c->persist = env_int("MI_TILE_PERSIST", 0);
    c->xcdRows = env_int("MI_XCD_ROWS", 1);
    c->deferPsi = env_int("MI_PCG_DEFER_PSI", 1);
    c->fusePerm = env_int("MI_FUSE_PERM", 1);
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess) c->nCU = pr.multiProcessorCount; }
    c->pairAT = env_int("MI_PBICG_PAIR", 1);
    c->fusePrologue = env_int("MI_FUSE_PROLOGUE", 1);
    c->multiPipe = env_int("MI_MULTI_PIPE", 1);   // the multi-vector tile passes of the Krylov iterations as persistent pipelined workgroups (multi_pipe.inc): same bits, three-component PBiCG + DILU iteration 1 812 -> 1 643 us (profiles/r05_f_multi_pipe_ab.md)
    c->winDirect = env_int("MI_WIN_DIRECT", 1); c->gamgGraphAttached = env_int("MI_GAMG_GRAPH_ATTACHED", 1);
    c->pcgPersist = env_int("MI_PCG_PERSIST", 1);
    c->pcgFuseRP = env_int("MI_PCG_FUSE_RP", 1) != 0 ? -1 : 0;
    c->persistGrid = env_int("MI_PERSIST_GRID", 0);
    c->persistShared = env_int("MI_PERSIST_SHARED", 0);
    c->fuseFinal = env_int("MI_PCG_FUSE_FINAL", 0); // measured: no gain (332.0 vs 332.3 us/iter), kept as an option
    c->amulBS = env_int("MI_AMUL_BS", 0); // 0 = choose per launch from the LDS footprint
    c->pcgBatch = env_int("MI_PCG_BATCH", 16); c->pcgGraph = env_int("MI_PCG_GRAPH", -1); c->pbicgHostStepped = env_int("MI_PBICG_HOST_STEPPED", 0);
    c->gamgDeviceInvert = env_int("MI_GAMG_DEVICE_INVERT", -1); c->gamgAlwaysAgglomerate = env_int("MI_GAMG_ALWAYS_AGGLOMERATE", 0); c->gamgGraph = env_int("MI_GAMG_GRAPH", 1); c->gamgFuse = env_int("MI_GAMG_FUSE", 1);
    if (c->amulBS != 256 && c->amulBS != 512 && c->amulBS != 1024) c->amulBS = 0;
    *out = c;
    return MI_OK;
}

extern "C" int mi_ctx_destroy(mi_ctx_t c)
{
    if (!c) return MI_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->hostState) (void)hipHostFree(c->hostState);
    if (c->hostScal) (void)hipHostFree(c->hostScal);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->sideStream) { (void)hipStreamSynchronize(c->sideStream); (void)hipStreamDestroy(c->sideStream); }
    if (c->evSideGo) (void)hipEventDestroy(c->evSideGo);
    if (c->evSideDone) (void)hipEventDestroy(c->evSideDone);
    if (c->ownStream) (void)hipStreamDestroy(c->stream);
    delete c;
    return MI_OK;
}

extern "C" int mi_ctx_synchronize(mi_ctx_t c)
{
    if (!c) return fail(MI_ERR_ARG, "ctx is NULL");
    HIPCHK(hipStreamSynchronize(c->stream));
    return MI_OK;
}

// run-time switches of a context (the environment variables of the same meaning are read once, at mi_ctx_create):
//   "pcg_persist"  0 / 1: the persistent PCG kernel for matrices that fit the CUs' registers (MI_PCG_PERSIST)
//   "fuse_prologue" 0 / 1: the solvers' prologue as one tile pass (A psi, source - A psi, sumA) + batched sums (MI_FUSE_PROLOGUE)
//   "win_direct"   0 / 1: boundary tiles of attached matrices read the halo window themselves (MI_WIN_DIRECT)
//   "gamg_graph_attached" 0 / 1: hipGraph replay of the V-cycle of a decomposed case (MI_GAMG_GRAPH_ATTACHED)
extern "C" int mi_ctx_set_option(mi_ctx_t c, const char* name, int32_t value)
{
    if (!c || !name) return fail(MI_ERR_ARG, "mi_ctx_set_option: bad argument");
    if (std::string(name) == "pcg_persist") { c->pcgPersist = value; return MI_OK; }
    if (std::string(name) == "pcg_fuse_rp") { c->pcgFuseRP = value != 0 ? -1 : 0; return MI_OK; }
    if (std::string(name) == "pcg_fuse_test") { c->pcgFuseTest = value; return MI_OK; }
    if (std::string(name) == "fuse_prologue") { c->fusePrologue = value; return MI_OK; }             // MI_FUSE_PROLOGUE
    if (std::string(name) == "win_direct") { c->winDirect = value; return MI_OK; }                     // MI_WIN_DIRECT
    if (std::string(name) == "gamg_graph_attached") { c->gamgGraphAttached = value; return MI_OK; }   // MI_GAMG_GRAPH_ATTACHED
    return fail(MI_ERR_ARG, "mi_ctx_set_option: unknown option");
}

// which solver paths ran on this context: 0 / 1 = launches of the persistent PCG kernel (one per batch of iterations) on plain /
// attached matrices, 2 = grid-barrier litmus runs (persist.inc), 3 = V-cycles of a decomposed case replayed as a hipGraph
extern "C" int mi_ctx_stat(mi_ctx_t c, int32_t which, int64_t* out)
{
    if (!c || !out || which < 0 || which >= 5) return fail(MI_ERR_ARG, "mi_ctx_stat: bad argument");
    *out = c->stats[which];
    return MI_OK;
}

// ---------------------------------------------------------------------------
// addressing
// ---------------------------------------------------------------------------
extern "C" int mi_addr_create_coupled(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces,
                                      const int32_t* lower, const int32_t* upper, int32_t n_patches,
                                      const int32_t* patch_sizes, const int32_t* const* patch_face_cells,
                                      const int32_t* const* patch_nbr_cells, mi_addr_t* out);

extern "C" int mi_addr_create(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces,
                              const int32_t* lower, const int32_t* upper, int32_t n_patches,
                              const int32_t* patch_sizes, const int32_t* const* patch_face_cells,
                              mi_addr_t* out)
{
    return mi_addr_create_coupled(ctx, n_cells, n_faces, lower, upper, n_patches, patch_sizes, patch_face_cells, nullptr, out);
}

static int addr_create_impl(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces, const int32_t* lower, const int32_t* upper, int32_t n_patches,
                            const int32_t* patch_sizes, const int32_t* const* patch_face_cells, const int32_t* const* patch_nbr_cells,
                            bool ordered, int32_t n_tiles, const int32_t* tile_cell_start, mi_addr_t* out, TileLayout* prebuilt);

extern "C" int mi_addr_create_coupled(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces,
                                      const int32_t* lower, const int32_t* upper, int32_t n_patches,
                                      const int32_t* patch_sizes, const int32_t* const* patch_face_cells,
                                      const int32_t* const* patch_nbr_cells, mi_addr_t* out)
{
    return addr_create_impl(ctx, n_cells, n_faces, lower, upper, n_patches, patch_sizes, patch_face_cells, patch_nbr_cells, false, 0, nullptr, out, nullptr);
}

// ORDERED addressing: the caller's numbering is kept (engine order == caller order, mi_addr_cell_perm is the identity), so the
// caller-order operators pay no permutation passes.  tile_cell_start (n_tiles + 1 offsets, from mi_addr_tile_starts of the
// layout whose mi_addr_cell_perm renumbered the mesh) names the tiles; NULL: consecutive cells are cut into tiles greedily.
extern "C" int mi_addr_create_ordered(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces, const int32_t* lower, const int32_t* upper,
                                      int32_t n_patches, const int32_t* patch_sizes, const int32_t* const* patch_face_cells,
                                      const int32_t* const* patch_nbr_cells, int32_t n_tiles, const int32_t* tile_cell_start, mi_addr_t* out)
{
    if (tile_cell_start && n_tiles <= 0) return fail(MI_ERR_ARG, "mi_addr_create_ordered: n_tiles must be positive when tile_cell_start is given");
    return addr_create_impl(ctx, n_cells, n_faces, lower, upper, n_patches, patch_sizes, patch_face_cells, patch_nbr_cells, true, n_tiles, tile_cell_start, out, nullptr);
}

// Renumber-at-bind (round 3): the mesh adopts the engine's cell order for its lifetime.  Host part (no device): the clustered layout
// of the mesh as given -> new-to-old cell map (the engine order) -> faces re-pointed, flipped where owner > neighbour, and sorted
// into upper-triangular order of the new numbering -- what polyMesh::renumber / renumberMesh.C do with a manual cell map.
namespace {
struct AdoptedMesh {
    Table<int32_t> cellMap, faceMap, lower, upper, tileStart;   // new -> old cell / face; addressing of the renumbered mesh
    Table<uint8_t> flipped;                                      // new face f has owner and neighbour swapped w.r.t. old face faceMap[f]
    Table<Table<int32_t>> patchFaceCells, patchNbrCells;   // renumbered
};
std::string adopt_engine_order(int32_t n_cells, int32_t n_faces, const int32_t* lower, const int32_t* upper, int32_t n_patches, const int32_t* patch_sizes,
                               const int32_t* const* patch_face_cells, const int32_t* const* patch_nbr_cells, AdoptedMesh& M)
{
    TileLayout L;
    TileParams prm;
    prm.tileCells = env_int("MI_TILE_CELLS", 0) > 0 ? env_int("MI_TILE_CELLS", 0) : 1024;
    prm.slotCap = env_int("MI_TILE_SLOTS", 4094);
    prm.reorder = env_int("MI_TILE_REORDER", -1);
    prm.compact = false;
    const std::string err = build_tile_layout(n_cells, n_faces, lower, upper, n_patches, patch_sizes, patch_face_cells, prm, L, patch_nbr_cells);
    if (!err.empty()) return err;
    M.cellMap = L.e2c; M.tileStart = L.tileCellStart;
    const Table<int32_t>& o2n = L.c2e;
    Table<int32_t> lo((size_t)n_faces), up((size_t)n_faces);
    Table<uint8_t> fl((size_t)n_faces);
    for (int32_t f = 0; f < n_faces; ++f) {
        const int32_t a = o2n[(size_t)lower[f]], b = o2n[(size_t)upper[f]];
        fl[(size_t)f] = a > b; lo[(size_t)f] = a < b ? a : b; up[(size_t)f] = a < b ? b : a;
    }
    // owner-sorted, then by neighbour, ties in old face order (two counting-sort passes: stable)
    Table<int32_t> byUp((size_t)n_faces), cnt((size_t)n_cells + 1, 0);
    for (int32_t f = 0; f < n_faces; ++f) cnt[(size_t)up[(size_t)f] + 1]++;
    for (int32_t c = 0; c < n_cells; ++c) cnt[(size_t)c + 1] += cnt[(size_t)c];
    for (int32_t f = 0; f < n_faces; ++f) byUp[(size_t)cnt[(size_t)up[(size_t)f]]++] = f;
    std::fill(cnt.begin(), cnt.end(), 0);
    for (int32_t f = 0; f < n_faces; ++f) cnt[(size_t)lo[(size_t)f] + 1]++;
    for (int32_t c = 0; c < n_cells; ++c) cnt[(size_t)c + 1] += cnt[(size_t)c];
    M.faceMap.resize((size_t)n_faces);
    for (int32_t k = 0; k < n_faces; ++k) { const int32_t f = byUp[(size_t)k]; M.faceMap[(size_t)cnt[(size_t)lo[(size_t)f]]++] = f; }
    M.lower.resize((size_t)n_faces); M.upper.resize((size_t)n_faces); M.flipped.resize((size_t)n_faces);
    for (int32_t k = 0; k < n_faces; ++k) { const int32_t f = M.faceMap[(size_t)k]; M.lower[(size_t)k] = lo[(size_t)f]; M.upper[(size_t)k] = up[(size_t)f]; M.flipped[(size_t)k] = fl[(size_t)f]; }
    M.patchFaceCells.resize((size_t)n_patches); M.patchNbrCells.resize((size_t)n_patches);
    for (int32_t p = 0; p < n_patches; ++p) {
        for (int32_t i = 0; i < patch_sizes[p]; ++i) M.patchFaceCells[(size_t)p].push_back(o2n[(size_t)patch_face_cells[p][i]]);
        if (patch_nbr_cells && patch_nbr_cells[p]) for (int32_t i = 0; i < patch_sizes[p]; ++i) M.patchNbrCells[(size_t)p].push_back(o2n[(size_t)patch_nbr_cells[p][i]]);
    }
    return std::string();
}
void adopted_maps_out(const AdoptedMesh& M, int32_t* cell_map, int32_t* face_map, uint8_t* flipped, int32_t* lower_out, int32_t* upper_out)
{
    if (cell_map) memcpy(cell_map, M.cellMap.data(), sizeof(int32_t) * M.cellMap.size());
    if (face_map) memcpy(face_map, M.faceMap.data(), sizeof(int32_t) * M.faceMap.size());
    if (flipped) memcpy(flipped, M.flipped.data(), M.flipped.size());
    if (lower_out) memcpy(lower_out, M.lower.data(), sizeof(int32_t) * M.lower.size());
    if (upper_out) memcpy(upper_out, M.upper.data(), sizeof(int32_t) * M.upper.size());
}
} // namespace

// host only (CPU tests): the renumbered mesh and its maps
extern "C" int mi_layout_adopt_host(int32_t n_cells, int32_t n_faces, const int32_t* lower, const int32_t* upper, int32_t n_patches, const int32_t* patch_sizes,
                                    const int32_t* const* patch_face_cells, const int32_t* const* patch_nbr_cells, int32_t* cell_new_to_old_out,
                                    int32_t* face_new_to_old_out, uint8_t* face_flipped_out, int32_t* lower_out, int32_t* upper_out, int32_t* n_tiles_out)
{
    if ((n_faces > 0 && (!lower || !upper)) || n_patches < 0) return fail(MI_ERR_ARG, "mi_layout_adopt_host: bad argument");
    AdoptedMesh M;
    const std::string err = adopt_engine_order(n_cells, n_faces, lower, upper, n_patches, patch_sizes, patch_face_cells, patch_nbr_cells, M);
    if (!err.empty()) return fail(MI_ERR_LIMIT, "mi_layout_adopt_host: " + err);
    adopted_maps_out(M, cell_new_to_old_out, face_new_to_old_out, face_flipped_out, lower_out, upper_out);
    if (n_tiles_out) *n_tiles_out = (int32_t)M.tileStart.size() - 1;
    return MI_OK;
}
extern "C" int mi_addr_create_adopted(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces, const int32_t* lower, const int32_t* upper, int32_t n_patches,
                                      const int32_t* patch_sizes, const int32_t* const* patch_face_cells, const int32_t* const* patch_nbr_cells,
                                      int32_t* cell_new_to_old_out, int32_t* face_new_to_old_out, uint8_t* face_flipped_out, int32_t* lower_out, int32_t* upper_out,
                                      mi_addr_t* out)
{
    if (!ctx || !out || (n_faces > 0 && (!lower || !upper)) || n_patches < 0) return fail(MI_ERR_ARG, "mi_addr_create_adopted: bad argument");
    AdoptedMesh M;
    const std::string err = adopt_engine_order(n_cells, n_faces, lower, upper, n_patches, patch_sizes, patch_face_cells, patch_nbr_cells, M);
    if (!err.empty()) return fail(MI_ERR_LIMIT, "mi_addr_create_adopted: " + err);
    Table<const int32_t*> pfc((size_t)n_patches), pnb((size_t)n_patches, nullptr);
    bool anyNbr = false;
    for (int32_t p = 0; p < n_patches; ++p) {
        pfc[(size_t)p] = M.patchFaceCells[(size_t)p].data();
        if (patch_nbr_cells && patch_nbr_cells[p]) { pnb[(size_t)p] = M.patchNbrCells[(size_t)p].data(); anyNbr = true; }
    }
    MICHK(addr_create_impl(ctx, n_cells, n_faces, M.lower.data(), M.upper.data(), n_patches, patch_sizes, pfc.data(), anyNbr ? pnb.data() : nullptr, true,
                           (int32_t)M.tileStart.size() - 1, M.tileStart.data(), out, nullptr));
    adopted_maps_out(M, cell_new_to_old_out, face_new_to_old_out, face_flipped_out, lower_out, upper_out);
    return MI_OK;
}

extern "C" int mi_addr_tile_starts(mi_addr_t a, int32_t* tile_cell_start_out)
{
    if (!a || !tile_cell_start_out) return fail(MI_ERR_ARG, "mi_addr_tile_starts: bad argument");
    memcpy(tile_cell_start_out, a->L.tileCellStart.data(), sizeof(int32_t) * a->L.tileCellStart.size());
    return MI_OK;
}
extern "C" int mi_addr_is_ordered(mi_addr_t a) { return a && a->identity ? 1 : 0; }

// the tile-layout parameters of an addressing of n_cells cells on this context (coarse: a level of a GAMG hierarchy)
static TileParams addr_tile_params(mi_ctx_t ctx, int32_t n_cells, bool coarse, bool ordered, int32_t n_tiles, const int32_t* tile_cell_start);
// prebuilt != nullptr: the host layout was built elsewhere (the GAMG builder builds its levels' layouts on other threads while
// it matches the next level) with addr_tile_params' parameters; it is moved from
static int addr_create_impl(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces, const int32_t* lower, const int32_t* upper, int32_t n_patches,
                            const int32_t* patch_sizes, const int32_t* const* patch_face_cells, const int32_t* const* patch_nbr_cells,
                            bool ordered, int32_t n_tiles, const int32_t* tile_cell_start, mi_addr_t* out, TileLayout* prebuilt);
static TileParams addr_tile_params(mi_ctx_t ctx, int32_t n_cells, bool coarse, bool ordered, int32_t n_tiles, const int32_t* tile_cell_start)
{
    TileParams prm;
    prm.tileCells = env_int("MI_TILE_CELLS", 0);
    if (prm.tileCells <= 0) {
        const int target = (coarse && env_int("MI_SMALL_TILES", 1)) ? (int)(((int64_t)n_cells / std::max(1, ctx->nCU) + 63) / 64 * 64) : 1024;
        prm.tileCells = std::min(1024, std::max(128, target));
    }
    prm.slotCap = env_int("MI_TILE_SLOTS", 4094);
    prm.reorder = env_int("MI_TILE_REORDER", -1);
    prm.compact = env_int("MI_ENTRY16", 0) != 0;
    prm.keepOrder = ordered; prm.givenTileStart = tile_cell_start; prm.nGivenTiles = n_tiles;
    return prm;
}
static int addr_create_impl(mi_ctx_t ctx, int32_t n_cells, int32_t n_faces, const int32_t* lower, const int32_t* upper, int32_t n_patches,
                            const int32_t* patch_sizes, const int32_t* const* patch_face_cells, const int32_t* const* patch_nbr_cells,
                            bool ordered, int32_t n_tiles, const int32_t* tile_cell_start, mi_addr_t* out, TileLayout* prebuilt)
{
    if (!ctx || !out || (n_faces > 0 && (!lower || !upper)) || n_patches < 0)
        return fail(MI_ERR_ARG, "mi_addr_create: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    mi_addr_s* a = new mi_addr_s();
    a->ctx = ctx;
    TileParams prm;
    prm.tileCells = env_int("MI_TILE_CELLS", 0);
    if (prm.tileCells <= 0) {
        // 1024 cells per tile; the coarse levels of a GAMG hierarchy (ctx->coarseLevelBuild) are cut into enough tiles for
        // every CU instead -- a shorter staging / row chain per workgroup, down to 128-cell tiles: their tile kernels take
        // 4.6-5.5 us instead of 7-8 (profiles/r02_gamg_rocprof_summary.md), 2.08 -> 2.02 ms per V-cycle on the 216^3 box.
        // The caller's own matrices keep 1024 (no gain measured for small PCG cases, and the partial-sum grouping of the
        // Krylov reductions stays what the parity tests pinned).
        const int target = (ctx->coarseLevelBuild && env_int("MI_SMALL_TILES", 1)) ? (int)(((int64_t)n_cells / std::max(1, ctx->nCU) + 63) / 64 * 64) : 1024;
        prm.tileCells = std::min(1024, std::max(128, target));
    }
    prm.slotCap = env_int("MI_TILE_SLOTS", 4094);
    prm.reorder = env_int("MI_TILE_REORDER", -1); // -1: Cuthill-McKee pre-ordering when the numbering has no locality (tiling.hpp)
    prm.compact = env_int("MI_ENTRY16", 0) != 0; // opt-in: half the entry bytes, measured 2-4 % slower (profiles/r01_n_compact_entries_ab.md)
    prm.keepOrder = ordered; prm.givenTileStart = tile_cell_start; prm.nGivenTiles = n_tiles;
#ifdef MI_TIMING
    auto ta__ = std::chrono::steady_clock::now();
    auto ta_tick = [&](const char* what) { auto n = std::chrono::steady_clock::now(); if (n_cells > 1000000) fprintf(stderr, "[addr] %-28s %.4f s\n", what, std::chrono::duration<double>(n - ta__).count()); ta__ = n; };
#endif
    if (prebuilt) a->L = std::move(*prebuilt);
    else {
    const std::string err = build_tile_layout(n_cells, n_faces, lower, upper, n_patches, patch_sizes, patch_face_cells, prm, a->L, patch_nbr_cells);
    if (!err.empty()) { delete a; return fail(MI_ERR_LIMIT, "mi_addr_create: " + err); }
    }
#ifdef MI_TIMING
    ta_tick("tile layout (host)");
#endif
    a->identity = true;
    for (int32_t e = 0; e < n_cells; ++e) if (a->L.e2c[(size_t)e] != e) { a->identity = false; break; }
    a->nLocalPatches = 0;
    a->patchIsLocal.assign((size_t)std::max(n_patches, 0), 0);
    if (patch_nbr_cells) for (int32_t p = 0; p < n_patches; ++p) if (patch_nbr_cells[p]) { a->nLocalPatches++; a->patchIsLocal[(size_t)p] = 1; }
    TileLayout& L = a->L;
    hipStream_t s = ctx->stream;
    int r = MI_OK;
#define UP(buf, vec) if (r == MI_OK) r = a->buf.upload(vec, s)
    UP(e2c, L.e2c); UP(c2e, L.c2e); UP(tileCellStart, L.tileCellStart); UP(tileSlotStart, L.tileSlotStart); UP(tileIfaceSlot0, L.tileIfaceSlot0);
    UP(tileHaloStart, L.tileHaloStart); UP(haloCell, L.haloCell); UP(tileSliceStart, L.tileSliceStart);
    a->compact = L.compact;
    if (L.compact) { UP(sliceEntryStart16, L.sliceEntryStart16); UP(entries16, L.entries16); UP(slotBase, L.slotBase); UP(tileSbStart, L.tileSbStart); }
    else { UP(sliceEntryStart, L.sliceEntryStart); UP(entries, L.entries); }
    UP(slotFace, L.slotFace);
    UP(extSlot, L.extSlot); UP(interiorTiles, L.interiorTiles); UP(boundaryTiles, L.boundaryTiles);
    UP(patchFaceCellsE, L.patchFaceCellsE); UP(faceSlot, L.faceSlot);
#undef UP
    if (r != MI_OK) { delete a; return r; }
    if (hipStreamSynchronize(s) != hipSuccess) { delete a; return fail(MI_ERR_DEVICE, "upload failed"); }
#ifdef MI_TIMING
    ta_tick("uploads + synchronise");
#endif
    a->nInterior = (int32_t)L.interiorTiles.size();
    a->nBoundary = (int32_t)L.boundaryTiles.size();
    a->nEntries = (int64_t)(L.compact ? L.entries16.size() : L.entries.size()); // 32-bit words of row entries on the device
    a->nHaloTot = (int64_t)L.haloCell.size();
    a->lowerHost.resize((size_t)n_faces); a->upperHost.resize((size_t)n_faces);
    mi::parallel_blocks(n_faces, 1 << 20, [&](int64_t b, int64_t e, int) {
        std::memcpy(a->lowerHost.data() + b, lower + b, sizeof(int32_t) * (size_t)(e - b));
        std::memcpy(a->upperHost.data() + b, upper + b, sizeof(int32_t) * (size_t)(e - b));
    });
    a->patchFaceCellsHost.resize((size_t)n_patches); a->patchNbrCellsHost.resize((size_t)n_patches);
    for (int32_t p = 0; p < n_patches; ++p) {
        a->patchFaceCellsHost[(size_t)p].assign(patch_face_cells[p], patch_face_cells[p] + patch_sizes[p]);
        if (a->patchIsLocal[(size_t)p]) a->patchNbrCellsHost[(size_t)p].assign(patch_nbr_cells[p], patch_nbr_cells[p] + patch_sizes[p]);
    }
    // drop the big host tables that only the device needs (their memory goes back to the system on another thread)
    {
        Table<int32_t> slotFace, faceSlot;
        if (!ctx->keepSlotTables) { slotFace.swap(L.slotFace); faceSlot.swap(L.faceSlot); }
        mi::free_in_background(L.entries, L.entries16, L.sliceEntryStart16, L.slotBase, slotFace, L.haloCell, L.sliceEntryStart, faceSlot);
        L.entries = Table<uint32_t>(); L.entries16 = Table<uint32_t>(); L.sliceEntryStart16 = Table<int32_t>(); L.slotBase = Table<uint16_t>();
        L.haloCell = Table<int32_t>(); L.sliceEntryStart = Table<int32_t>();
    }
#ifdef MI_TIMING
    ta_tick("host copies + frees");
#endif
    *out = a;
    return MI_OK;
}

extern "C" int mi_addr_destroy(mi_addr_t a) { delete a; return MI_OK; }

extern "C" int mi_addr_set_ami_patch(mi_addr_t a, int32_t patch, int32_t nbr_patch, const int32_t* start, const int32_t* address, const double* weights,
                                     const uint8_t* low_weight)
{
    if (!a || patch < 0 || patch >= a->L.nPatches || nbr_patch < 0 || nbr_patch >= a->L.nPatches)
        return fail(MI_ERR_ARG, "mi_addr_set_ami_patch: bad argument");
    if (a->patchIsLocal[(size_t)patch]) return fail(MI_ERR_ARG, "mi_addr_set_ami_patch: the patch must be created without neighbour cells (ext region), once");
    if ((start || address || weights) && !(start && address && weights)) return fail(MI_ERR_ARG, "mi_addr_set_ami_patch: start, address and weights go together");
    HIPCHK(hipSetDevice(a->ctx->device));
    const Table<int32_t>& mine = a->patchFaceCellsHost[(size_t)patch];
    const Table<int32_t>& theirs = a->patchFaceCellsHost[(size_t)nbr_patch];
    const int32_t n = (int32_t)mine.size(), nn = (int32_t)theirs.size();
    Table<int32_t> st((size_t)n + 1, 0), ce, own;
    Table<double> w;
    if (!start) { // one face to one face with unit weight: a cyclic patch that needs its transformation factor
        if (n != nn) return fail(MI_ERR_ARG, "mi_addr_set_ami_patch: one-to-one coupling needs patches of equal size");
        ce.resize((size_t)n); w.assign((size_t)n, 1.0);
        for (int32_t i = 0; i < n; ++i) { st[(size_t)i + 1] = i + 1; ce[i] = a->L.c2e[(size_t)theirs[i]]; }
    } else {
        if (start[0] != 0) return fail(MI_ERR_ARG, "mi_addr_set_ami_patch: start[0] must be 0");
        for (int32_t i = 0; i < n; ++i) if (start[i + 1] < start[i]) return fail(MI_ERR_ARG, "mi_addr_set_ami_patch: start must be non-decreasing");
        const int32_t na = start[n];
        ce.resize((size_t)na); w.assign(weights, weights + na);
        for (int32_t k = 0; k < na; ++k) {
            if (address[k] < 0 || address[k] >= nn) return fail(MI_ERR_ARG, "mi_addr_set_ami_patch: address outside the neighbour patch");
            ce[k] = a->L.c2e[(size_t)theirs[(size_t)address[k]]];
        }
        st.assign(start, start + n + 1);
    }
    mi_addr_s::AmiPatch* q = new mi_addr_s::AmiPatch();
    q->patch = patch; q->nbrPatch = nbr_patch; q->n = n; q->extOff = a->L.patchOffset[(size_t)patch];
    q->hStart = st; q->hW = w;
    if (start) q->hAddr.assign(address, address + start[n]); else { q->hAddr.resize((size_t)n); for (int32_t i = 0; i < n; ++i) q->hAddr[i] = i; }
    if (low_weight) {
        own.assign((size_t)n, -1);
        for (int32_t i = 0; i < n; ++i) if (low_weight[i]) { own[i] = a->L.c2e[(size_t)mine[i]]; q->hasLow = true; }
    }
    hipStream_t s = a->ctx->stream;
    int r = q->start.upload(st, s);
    if (r == MI_OK && !ce.empty()) r = q->cellE.upload(ce, s);
    if (r == MI_OK && !w.empty()) r = q->w.upload(w, s);
    if (r == MI_OK && q->hasLow) r = q->ownE.upload(own, s);
    if (r != MI_OK) { delete q; return r; }
    HIPCHK(hipStreamSynchronize(s));
    a->ami.push_back(q);
    a->patchIsLocal[(size_t)patch] = 2; // no exchange: the values come from this rank's own cells
    a->nLocalPatches++;
    return MI_OK;
}

// cyclicAMI whose partner patch lives on another rank (AMIInterpolation with singlePatchProc == -1, AMIInterpolation.C:940-1091:
// the reference ships the partner's field with a mapDistribute).  Here the partner's patch-internal field arrives through an
// ordinary PROCESSOR patch of this addressing, the TRANSPORT patch: same size on both ranks (the larger of the two AMI sides,
// padded), faceCells = this side's AMI faceCells (padding: any cell), zero interface coefficients -- so every transport the
// engine has (send / recv, peer windows, the external callbacks) carries it unchanged, and the matrix never sees it.  The
// interpolation then reads the transport patch's ext region: address[k] = index of the transport-patch face whose received
// value is the k-th contribution (finest level: the partner's face number).  Everything else -- weights, low-weight faces,
// transformCoupleField factor, agglomeration per GAMG level -- as mi_addr_set_ami_patch.
// the general form: the partner SIDE is split over several ranks (AMIInterpolation.C:940-1091 calcProcMap: what a rank needs arrives from
// all ranks and is numbered rank by rank) -- one transport patch per partner piece; address k names a received value as (slot of the
// transport in `transports`, face of that transport patch).  hAddrConcat: the same addresses in the concatenated numbering of the
// pieces (kept for the GAMG builder).
int ami_remote_impl(mi_addr_s* a, int32_t patch, const Table<int32_t>& transports, const Table<int32_t>& partCount, const int32_t* start,
                    const int32_t* addrSlot, const int32_t* addrFace, const int32_t* hAddrConcat, const double* weights, const uint8_t* low_weight)
{
    const char* me = "mi_addr_set_ami_patch_remote";
    if (!a || patch < 0 || patch >= a->L.nPatches || transports.empty() || transports.size() != partCount.size() || !start || !weights) return fail(MI_ERR_ARG, std::string(me) + ": bad argument");
    if (a->patchIsLocal[(size_t)patch]) return fail(MI_ERR_ARG, std::string(me) + ": the cyclicAMI patch must be created without neighbour cells (an ext region)");
    int32_t total = 0;
    for (size_t q = 0; q < transports.size(); ++q) {
        const int32_t t = transports[q];
        if (t < 0 || t >= a->L.nPatches || t == patch || a->patchIsLocal[(size_t)t]) return fail(MI_ERR_ARG, std::string(me) + ": a transport patch must be a processor patch of this addressing (created without neighbour cells)");
        for (size_t r = 0; r < q; ++r) if (transports[r] == t) return fail(MI_ERR_ARG, std::string(me) + ": a transport patch is listed twice");
        const int32_t nT = a->L.patchOffset[(size_t)t + 1] - a->L.patchOffset[(size_t)t];
        if (partCount[q] < 0 || partCount[q] > nT) return fail(MI_ERR_ARG, std::string(me) + ": a transport patch is smaller than the partner piece it carries");
        total += partCount[q];
    }
    HIPCHK(hipSetDevice(a->ctx->device));
    const Table<int32_t>& mine = a->patchFaceCellsHost[(size_t)patch];
    const int32_t n = (int32_t)mine.size();
    if (start[0] != 0) return fail(MI_ERR_ARG, std::string(me) + ": start[0] must be 0");
    for (int32_t i = 0; i < n; ++i) if (start[i + 1] < start[i]) return fail(MI_ERR_ARG, std::string(me) + ": start must be non-decreasing");
    const int32_t na = start[n];
    Table<int32_t> st(start, start + n + 1), ce((size_t)na), own;
    for (int32_t k = 0; k < na; ++k) {
        const int32_t q = addrSlot[k];
        if (q < 0 || q >= (int32_t)transports.size() || addrFace[k] < 0 || addrFace[k] >= a->L.patchOffset[(size_t)transports[(size_t)q] + 1] - a->L.patchOffset[(size_t)transports[(size_t)q]])
            return fail(MI_ERR_ARG, std::string(me) + ": address outside its transport patch");
        ce[(size_t)k] = a->L.nCells + a->L.patchOffset[(size_t)transports[(size_t)q]] + addrFace[k];   // engine vectors: ext region behind the owned cells
    }
    mi_addr_s::AmiPatch* q = new mi_addr_s::AmiPatch();
    q->patch = patch; q->nbrPatch = transports[0]; q->n = n; q->extOff = a->L.patchOffset[(size_t)patch];
    q->transport = transports[0]; q->nPartner = total; q->transports = transports; q->partCount = partCount;
    q->hStart = st; q->hW.assign(weights, weights + na); q->hAddr.assign(hAddrConcat, hAddrConcat + na);
    if (low_weight) {
        own.assign((size_t)n, -1);
        for (int32_t i = 0; i < n; ++i) if (low_weight[i]) { own[i] = a->L.c2e[(size_t)mine[i]]; q->hasLow = true; }
    }
    hipStream_t s = a->ctx->stream;
    int r = q->start.upload(st, s);
    if (r == MI_OK && !ce.empty()) r = q->cellE.upload(ce, s);
    if (r == MI_OK && na > 0) r = q->w.upload(q->hW, s);
    if (r == MI_OK && q->hasLow) r = q->ownE.upload(own, s);
    if (r != MI_OK) { delete q; return r; }
    HIPCHK(hipStreamSynchronize(s));
    a->ami.push_back(q);
    a->patchIsLocal[(size_t)patch] = 2;   // no exchange for the AMI patch itself: its values are interpolated from the transport patches'
    a->nLocalPatches++;
    a->amiRemote = true;
    return MI_OK;
}
// addresses in the concatenated numbering of the partner pieces: piece q = faces [sum of n_partner_faces[0..q), + n_partner_faces[q]),
// its face j travelling as face j of transport patch q
extern "C" int mi_addr_set_ami_patch_remote_multi(mi_addr_t a, int32_t patch, int32_t n_transports, const int32_t* transport_patches, const int32_t* n_partner_faces,
                                                  const int32_t* start, const int32_t* address, const double* weights, const uint8_t* low_weight)
{
    if (!a || patch < 0 || patch >= a->L.nPatches || n_transports < 1 || !transport_patches || !n_partner_faces || !start || !address || !weights)
        return fail(MI_ERR_ARG, "mi_addr_set_ami_patch_remote_multi: bad argument");
    const int32_t n = (int32_t)a->patchFaceCellsHost[(size_t)patch].size();
    Table<int32_t> tr(transport_patches, transport_patches + n_transports), cnt(n_partner_faces, n_partner_faces + n_transports), off((size_t)n_transports + 1, 0);
    for (int32_t q = 0; q < n_transports; ++q) { if (cnt[(size_t)q] < 0) return fail(MI_ERR_ARG, "mi_addr_set_ami_patch_remote_multi: negative partner face count"); off[(size_t)q + 1] = off[(size_t)q] + cnt[(size_t)q]; }
    for (int32_t i = 0; i < n; ++i) if (start[i + 1] < start[i]) return fail(MI_ERR_ARG, "mi_addr_set_ami_patch_remote_multi: start must be non-decreasing");
    const int32_t na = n > 0 || start ? start[n] : 0;
    Table<int32_t> slot((size_t)na), face((size_t)na);
    for (int32_t k = 0; k < na; ++k) {
        if (address[k] < 0 || address[k] >= off[(size_t)n_transports]) return fail(MI_ERR_ARG, "mi_addr_set_ami_patch_remote_multi: address outside the partner pieces");
        int32_t q = 0;
        while (address[k] >= off[(size_t)q + 1]) ++q;
        slot[(size_t)k] = q; face[(size_t)k] = address[k] - off[(size_t)q];
    }
    return ami_remote_impl(a, patch, tr, cnt, start, slot.data(), face.data(), address, weights, low_weight);
}
extern "C" int mi_addr_set_ami_patch_remote(mi_addr_t a, int32_t patch, int32_t transport_patch, int32_t n_partner_faces, const int32_t* start,
                                            const int32_t* address, const double* weights, const uint8_t* low_weight)
{
    return mi_addr_set_ami_patch_remote_multi(a, patch, 1, &transport_patch, &n_partner_faces, start, address, weights, low_weight);
}

extern "C" int mi_addr_set_ami_face_areas(mi_addr_t a, int32_t patch, const double* mag_sf)
{
    if (!a || !mag_sf) return fail(MI_ERR_ARG, "mi_addr_set_ami_face_areas: bad argument");
    for (mi_addr_s::AmiPatch* q : a->ami) if (q->patch == patch) { q->hMagSf.assign(mag_sf, mag_sf + q->n); return MI_OK; }
    return fail(MI_ERR_ARG, "mi_addr_set_ami_face_areas: not a cyclicAMI patch");
}

extern "C" int mi_matrix_set_patch_transform(mi_matrix_t m, int32_t patch, double factor)
{
    if (!m || patch < 0 || patch >= m->addr->L.nPatches) return fail(MI_ERR_ARG, "mi_matrix_set_patch_transform: bad argument");
    if (m->addr->patchIsLocal[(size_t)patch] == 1 && factor != 1.0)
        return fail(MI_ERR_UNSUPPORTED, "mi_matrix_set_patch_transform: a transformed cyclic patch is declared through mi_addr_set_ami_patch (one-to-one, unit weights)");
    // the ranks of a decomposed case agreed on their solver pipelines when the matrix was attached (comm_any_factor)
    if (comm_attached(m) && factor != 1.0 && !comm_any_factor(m))
        return fail(MI_ERR_STATE, "mi_matrix_set_patch_transform: the first transformed patch of a decomposed case is declared BEFORE mi_matrix_attach_comm (the ranks agree on the solver pipeline there)");
    if (m->patchFactor.empty()) m->patchFactor.assign((size_t)m->addr->L.nPatches, 1.0);
    if (m->patchFactor[(size_t)patch] != factor) ++m->epoch; // a GAMG hierarchy keeps the factors with its level matrices: rebuild them
    m->patchFactor[(size_t)patch] = factor;
    return MI_OK;
}
extern "C" int32_t mi_addr_n_cells(mi_addr_t a) { return a ? a->L.nCells : 0; }
extern "C" int32_t mi_addr_n_faces(mi_addr_t a) { return a ? a->L.nFaces : 0; }
extern "C" int32_t mi_addr_n_tiles(mi_addr_t a) { return a ? a->L.nTiles : 0; }
extern "C" int32_t mi_addr_n_ext(mi_addr_t a) { return a ? a->L.nExt : 0; }

extern "C" int mi_addr_cell_perm(mi_addr_t a, int32_t* e2c_host)
{
    if (!a || !e2c_host) return fail(MI_ERR_ARG, "mi_addr_cell_perm: bad argument");
    memcpy(e2c_host, a->L.e2c.data(), sizeof(int32_t) * (size_t)a->L.nCells);
    return MI_OK;
}

extern "C" int mi_addr_patch_offsets(mi_addr_t a, int32_t* off)
{
    if (!a || !off) return fail(MI_ERR_ARG, "mi_addr_patch_offsets: bad argument");
    memcpy(off, a->L.patchOffset.data(), sizeof(int32_t) * a->L.patchOffset.size());
    return MI_OK;
}

static size_t lds_bytes(const TileLayout& L, bool asym, bool ainv, int32_t* offLow, int32_t* offX, int32_t* offRD, int32_t* offSB = nullptr)
{
    const int32_t slots = (L.maxSlots + 3) & ~1; // even
    const int32_t xlen = ((L.maxCells + 63) & ~63) + L.maxHalo + 2;
    int32_t off = slots;
    *offLow = off; if (asym) off += slots;
    *offX = off; off += (xlen + 1) & ~1;
    *offRD = off; if (ainv) off += (xlen + 1) & ~1;
    if (offSB) { *offSB = off; if (L.compact) off += (L.maxCells + L.maxHalo + 1 + 3) / 4 + 1; } // uint16 slot bases
    return (size_t)off * sizeof(double);
}

extern "C" int mi_addr_stats(mi_addr_t a, int64_t st[8])
{
    if (!a || !st) return fail(MI_ERR_ARG, "mi_addr_stats: bad argument");
    int32_t o1, o2, o3;
    st[0] = a->L.nTiles; st[1] = a->L.totalSlots; st[2] = a->nEntries; st[3] = a->nHaloTot;
    st[4] = a->L.maxCells; st[5] = a->L.maxSlots; st[6] = a->L.maxHalo;
    int32_t o4;
    st[7] = (int64_t)lds_bytes(a->L, false, false, &o1, &o2, &o3, &o4);
    return MI_OK;
}

// ---------------------------------------------------------------------------
// matrix
// ---------------------------------------------------------------------------
// process-wide stamp of a coefficient binding: (matrix, epoch) never repeats, even when a new matrix reuses an old address
static uint64_t next_epoch() { static std::atomic<uint64_t> e{0}; return ++e; }
extern "C" mi_addr_t mi_matrix_addr(mi_matrix_t m) { return m ? m->addr : nullptr; }
extern "C" int mi_matrix_create(mi_addr_t a, mi_matrix_t* out)
{
    if (!a || !out) return fail(MI_ERR_ARG, "mi_matrix_create: bad argument");
    HIPCHK(hipSetDevice(a->ctx->device));
    mi_matrix_s* m = new mi_matrix_s();
    m->addr = a;
    int r = m->diagE.alloc((size_t)a->L.nCells);
    if (r == MI_OK) r = m->upE.alloc((size_t)a->L.totalSlots);
    if (r != MI_OK) { delete m; return r; }
    if (hipMemsetAsync(m->upE.p, 0, sizeof(double) * (size_t)a->L.totalSlots, a->ctx->stream) != hipSuccess) { delete m; return fail(MI_ERR_DEVICE, "memset failed"); }
    *out = m;
    return MI_OK;
}

extern "C" int mi_matrix_destroy(mi_matrix_t m)
{
    if (m) { (void)hipSetDevice(m->addr->ctx->device); (void)hipStreamSynchronize(m->addr->ctx->stream); delete m; }
    return MI_OK;
}

extern "C" int mi_matrix_set_coeffs(mi_matrix_t m, const double* diag, const double* upper, const double* lower)
{
    if (!m || !diag || (!upper && m->addr->L.nFaces > 0)) return fail(MI_ERR_ARG, "mi_matrix_set_coeffs: bad argument");
    mi_addr_s* a = m->addr;
    hipStream_t s = a->ctx->stream;
    HIPCHK(hipSetDevice(a->ctx->device));
    const bool asym = (lower != nullptr);
    if (asym && m->lowE.n != (size_t)a->L.totalSlots) {
        MICHK(m->lowE.alloc((size_t)a->L.totalSlots));
        // interface slots of a previously symmetric matrix: lower side mirrors upper until re-set
        HIPCHK(hipMemcpyAsync(m->lowE.p, m->upE.p, sizeof(double) * (size_t)a->L.totalSlots, hipMemcpyDeviceToDevice, s));
    }
    m->asym = asym;
    k_gather_perm<<<RG, RB, 0, s>>>(diag, a->perm(), m->diagE.p, a->L.nCells);
    k_fill_slots<<<2048, 256, 0, s>>>(upper, lower, a->slotFace.p, m->upE.p, asym ? m->lowE.p : nullptr, a->L.totalSlots);
    HIPCHK(hipGetLastError());
    m->bound = true;
    m->rDValid = false; m->sumAValid = false;
    m->epoch = next_epoch();
    return MI_OK;
}

extern "C" int mi_matrix_set_interface_coeffs(mi_matrix_t m, int32_t patch, const double* bou, const double* inte)
{
    if (!m || !bou || patch < 0 || patch >= m->addr->L.nPatches) return fail(MI_ERR_ARG, "mi_matrix_set_interface_coeffs: bad argument");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    const int32_t off = a->L.patchOffset[patch], n = a->L.patchOffset[(size_t)patch + 1] - off;
    if (n == 0) return MI_OK;
    k_fill_iface<<<(n + 255) / 256, 256, 0, a->ctx->stream>>>(bou, inte, a->extSlot.p + off, m->upE.p, m->asym ? m->lowE.p : nullptr, n);
    HIPCHK(hipGetLastError());
    m->sumAValid = false;
    m->epoch = next_epoch();
    return MI_OK;
}

// ---------------------------------------------------------------------------
// tile kernel launch
// ---------------------------------------------------------------------------
namespace {

template <int OP, bool ASYM, bool TRANS, bool C16>
int launch_tile_bs(mi_matrix_s* m, const TileArgs& args, int nTiles, size_t lds)
{
    hipStream_t s = m->addr->ctx->stream;
    // residency: three 512-thread workgroups per CU while a tile needs <= 53 KiB of LDS; with more LDS
    // (asymmetric matrices, AINV) only two fit, and 1024 threads keep the CU at 32 waves (measured:
    // asymmetric Amul 254 -> 211 us on the 216^3 box)
    int bs = m->addr->ctx->amulBS;
    if (bs == 0) bs = (lds > 53 * 1024) ? 1024 : (m->addr->L.maxCells <= 256 ? 256 : 512); // small tiles (small matrices): four 64-row slices at most
    if (nTiles <= 0) return MI_OK;
#define MI_LAUNCH(BS)                                                                                                   \
    {                                                                                                                   \
        mi_ctx_s* cx = m->addr->ctx;                                                                                    \
        const void* fn = (const void*)tile_kernel<OP, ASYM, TRANS, BS, C16>;                                            \
        if (!cx->ldsAttrSet.count(fn)) { /* per context = per device: a process may drive several devices */          \
            HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));                    \
            cx->ldsAttrSet.insert(fn);                                                                                  \
        }                                                                                                               \
        int grid = nTiles;                                                                                              \
        if (cx->persist) {                                                                                              \
            int& occ = cx->occCache[std::make_pair(fn, lds)];                                                           \
            if (occ == 0) {                                                                                             \
                int nb = 0;                                                                                             \
                HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, BS, lds));                                 \
                occ = nb > 0 ? nb : 1;                                                                                  \
            }                                                                                                           \
            const int slots = ((occ * cx->nCU * cx->persist) / 8) * 8;                                                  \
            if (slots >= 8 && nTiles > 2 * slots) grid = slots;                                                         \
        }                                                                                                               \
        if (m->kevStart) /* start/stop events stamped by the kernel's own begin/end: the profiler's clock */           \
            hipExtLaunchKernelGGL((tile_kernel<OP, ASYM, TRANS, BS, C16>), dim3(grid), dim3(BS), (uint32_t)lds, s, m->kevStart, m->kevStop, 0, args); \
        else                                                                                                            \
        tile_kernel<OP, ASYM, TRANS, BS, C16><<<grid, BS, lds, s>>>(args);                                                   \
    }
    if (bs == 1024) MI_LAUNCH(1024)
    else if (bs == 512) MI_LAUNCH(512)
    else MI_LAUNCH(256)
#undef MI_LAUNCH
    HIPCHK(hipGetLastError());
    return MI_OK;
}

// all tiles of an attached matrix in ONE launch, boundary tiles reading the halo window (peer.inc: tile_kernel_win)
template <int OP>
int launch_tile_win(mi_matrix_s* m, bool trans, const double* x, const double* b, const double* rD, double* y,
                    double omega, double* dotPartial, double* dotPartial2);

// which: 0 all tiles, 1 interior only, 2 boundary only
template <int OP>
int launch_tile(mi_matrix_s* m, bool trans, const double* x, const double* b, const double* rD, double* y,
                double omega, int which, double* dotPartial = nullptr, double* dotPartial2 = nullptr)
{
    mi_addr_s* a = m->addr;
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound (mi_matrix_set_coeffs)");
    TileArgs t;
    t.tileCellStart = a->tileCellStart.p; t.tileSlotStart = a->tileSlotStart.p; t.tileIfaceSlot0 = a->tileIfaceSlot0.p; t.tileHaloStart = a->tileHaloStart.p;
    t.haloCell = a->haloCell.p; t.tileSliceStart = a->tileSliceStart.p; t.sliceEntryStart = a->sliceEntryStart.p;
    t.entries = a->entries.p; t.entries16 = a->entries16.p; t.sliceEntryStart16 = a->sliceEntryStart16.p; t.slotBase = reinterpret_cast<const uint32_t*>(a->slotBase.p); t.tileSbStart = a->tileSbStart.p;
    t.diag = m->diagE.p; t.up = m->upE.p; t.low = m->lowE.p;
    t.x = x; t.b = b; t.rD = rD; t.y = y; t.omega = omega; t.dotPartial = dotPartial; t.dotPartial2 = dotPartial2; t.flags = a->ctx->tileFlags;
    if (OP == OP_PROLOGUE) { t.y2 = m->proY2; t.y3 = m->proY3; }

    const size_t lds = lds_bytes(a->L, m->asym, OP == OP_AINV, &t.offLow, &t.offX, &t.offRD, &t.offSB);
    if (lds > 159 * 1024) return fail(MI_ERR_LIMIT, "tile needs more than 159 KiB of LDS");
    int nTiles = a->L.nTiles;
    t.tileList = nullptr;
    if (which == 1) { t.tileList = a->interiorTiles.p; nTiles = a->nInterior; }
    else if (which == 2) { t.tileList = a->boundaryTiles.p; nTiles = a->nBoundary; }
    t.nPos = nTiles;
    t.done = m->gateDone ? &a->ctx->state.p->done : nullptr;
    if (m->callerY) { // caller-order launch (caller_op): the engine vector x only supplies its ext tail
        if constexpr (OP == OP_AMUL || OP == OP_RESIDUAL || OP == OP_H || OP == OP_SUMA || OP == OP_H1) {
            t.perm = a->e2c.p; t.haloSrc = a->haloSrc.p; t.xExt = x ? x + a->L.nCells : nullptr;
            t.x = m->callerX; t.b = m->callerB; t.y = m->callerY;
            mi_ctx_s* cx = a->ctx;
            hipStream_t s = cx->stream;
            const bool big = lds > 53 * 1024;
#define MI_LAUNCH_PERM(ASYM, TRANS)                                                                                      \
            {                                                                                                           \
                const void* fn = big ? (const void*)tile_kernel_perm<OP, ASYM, TRANS, 1024> : (const void*)tile_kernel_perm<OP, ASYM, TRANS, 512>; \
                if (!cx->ldsAttrSet.count(fn)) { HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)); cx->ldsAttrSet.insert(fn); } \
                if (big) tile_kernel_perm<OP, ASYM, TRANS, 1024><<<nTiles, 1024, lds, s>>>(t);                           \
                else tile_kernel_perm<OP, ASYM, TRANS, 512><<<nTiles, 512, lds, s>>>(t);                                 \
            }
            if (nTiles > 0) {
                if (m->asym) { if (trans) MI_LAUNCH_PERM(true, true) else MI_LAUNCH_PERM(true, false) }
                else MI_LAUNCH_PERM(false, false)
            }
#undef MI_LAUNCH_PERM
            HIPCHK(hipGetLastError());
            return MI_OK;
        } else return fail(MI_ERR_STATE, "caller-order tile launch of an operator without a caller-order form");
    }
    if (a->compact) {
        if (m->asym) {
            if (trans) return launch_tile_bs<OP, true, true, true>(m, t, nTiles, lds);
            return launch_tile_bs<OP, true, false, true>(m, t, nTiles, lds);
        }
        return launch_tile_bs<OP, false, false, true>(m, t, nTiles, lds);
    }
    if (m->asym) {
        if (trans) return launch_tile_bs<OP, true, true, false>(m, t, nTiles, lds);
        return launch_tile_bs<OP, true, false, false>(m, t, nTiles, lds);
    }
    return launch_tile_bs<OP, false, false, false>(m, t, nTiles, lds);
}

// tile pass with the operand formed in the staging (tile_kernel_fx; kernels.hip.hpp: TileArgs::fx*): every tile, engine order, no
// coupled patches (the operand of a halo cell is formed from the same arrays as an own cell's), plain row entries
struct FxArgs {
    const int32_t* map = nullptr; const double* coarse = nullptr; double* out = nullptr;
    const double *field = nullptr, *acf = nullptr, *src = nullptr, *scal = nullptr, *add = nullptr;
    unsigned int* foldCounter = nullptr; double* foldOut = nullptr;
};
inline bool tile_fx_usable(const mi_matrix_s* m) { return m->bound && !m->addr->compact && m->addr->L.nExt == 0 && m->addr->ami.empty() && !m->callerY && !m->gateDone; }
template <int OP, int XMODE>
int launch_tile_fx(mi_matrix_s* m, const FxArgs& fx, const double* b, double* y, double omega, double* dotPartial = nullptr, double* dotPartial2 = nullptr)
{
    mi_addr_s* a = m->addr;
    if (!tile_fx_usable(m)) return fail(MI_ERR_STATE, "launch_tile_fx: matrix not eligible");
    TileArgs t;
    t.tileCellStart = a->tileCellStart.p; t.tileSlotStart = a->tileSlotStart.p; t.tileIfaceSlot0 = a->tileIfaceSlot0.p; t.tileHaloStart = a->tileHaloStart.p;
    t.haloCell = a->haloCell.p; t.tileSliceStart = a->tileSliceStart.p; t.sliceEntryStart = a->sliceEntryStart.p;
    t.entries = a->entries.p; t.entries16 = nullptr; t.sliceEntryStart16 = nullptr; t.slotBase = nullptr; t.tileSbStart = nullptr;
    t.diag = m->diagE.p; t.up = m->upE.p; t.low = m->lowE.p;
    t.x = nullptr; t.b = b; t.rD = nullptr; t.y = y; t.omega = omega; t.dotPartial = dotPartial; t.dotPartial2 = dotPartial2; t.flags = a->ctx->tileFlags;
    t.fxMap = fx.map; t.fxCoarse = fx.coarse; t.fxOut = fx.out; t.fxField = fx.field; t.fxAcf = fx.acf; t.fxSrc = fx.src; t.fxScal = fx.scal; t.fxAdd = fx.add;
    t.foldCounter = fx.foldCounter; t.foldOut = fx.foldOut;
    size_t lds = lds_bytes(a->L, m->asym, false, &t.offLow, &t.offX, &t.offRD, &t.offSB);
    if (lds > 159 * 1024) return fail(MI_ERR_LIMIT, "tile needs more than 159 KiB of LDS");
    if (fx.foldOut) lds = std::max(lds, (size_t)(2 * RG + 2 * (RB / 64) + 2) * sizeof(double));   // the folding workgroup's scratch
    const int nTiles = a->L.nTiles;
    t.tileList = nullptr; t.nPos = nTiles; t.done = nullptr;
    if (nTiles <= 0) return MI_OK;
    mi_ctx_s* cx = a->ctx;
    hipStream_t s = cx->stream;
    int bs = cx->amulBS;
    if (bs == 0) bs = (lds > 53 * 1024) ? 1024 : (a->L.maxCells <= 256 ? 256 : 512);
#define MI_LAUNCH_FX(ASYM, BS)                                                                                           \
    {                                                                                                                   \
        const void* fn = (const void*)tile_kernel_fx<OP, ASYM, BS, XMODE>;                                              \
        if (!cx->ldsAttrSet.count(fn)) { HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)); cx->ldsAttrSet.insert(fn); } \
        tile_kernel_fx<OP, ASYM, BS, XMODE><<<nTiles, BS, lds, s>>>(t);                                                 \
    }
    if (m->asym) { if (bs == 1024) MI_LAUNCH_FX(true, 1024) else if (bs == 512) MI_LAUNCH_FX(true, 512) else MI_LAUNCH_FX(true, 256) }
    else { if (bs == 1024) MI_LAUNCH_FX(false, 1024) else if (bs == 512) MI_LAUNCH_FX(false, 512) else MI_LAUNCH_FX(false, 256) }
#undef MI_LAUNCH_FX
    HIPCHK(hipGetLastError());
    return MI_OK;
}

int ensure_rD(mi_matrix_s* m)
{
    if (m->rDValid) return MI_OK;
    mi_addr_s* a = m->addr;
    const size_t need = (size_t)a->L.nCells + (size_t)a->L.nExt;
    if (m->rD.n != need) {
        MICHK(m->rD.alloc(need));
        HIPCHK(hipMemsetAsync(m->rD.p, 0, need * sizeof(double), a->ctx->stream));
    }
    k_recip<<<RG, RB, 0, a->ctx->stream>>>(m->rD.p, m->diagE.p, a->L.nCells);
    HIPCHK(hipGetLastError());
    m->rDValid = true;
    return MI_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// cyclicAMI neighbour values (cyclicAMIFvPatchField.C:195-224, cyclicAMIGAMGInterfaceField.C:97-130): the partner patch's
// internal values, transformed (transformCoupleField: *= factor), then interpolated with the AMI weights --
// AMIInterpolationF.H:62-105 with multiplyWeightedOp<plusEqOp>: out += w*f in address order, one fma per term under nvcc's
// default contraction.  A face below the low-weight threshold takes its own cell's value (the `pif` default).
__global__ void k_ami_fill(const double* __restrict__ x, const int32_t* __restrict__ start, const int32_t* __restrict__ cellE,
                           const double* __restrict__ w, const int32_t* __restrict__ ownE, double factor, double* __restrict__ ext, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (ownE && ownE[i] >= 0) { ext[i] = x[ownE[i]]; return; }
    double acc = 0.0;
    for (int k = start[i]; k < start[i + 1]; ++k) { const double t = factor * x[cellE[k]]; acc = fma(w[k], t, acc); }
    ext[i] = acc;
}
__global__ void k_scale_range(double* __restrict__ v, double factor, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] *= factor;
}
double patch_factor(const mi_matrix_s* m, int32_t p) { return m->patchFactor.empty() ? 1.0 : m->patchFactor[(size_t)p]; }
// x: engine-order vector; fills the ext values of every cyclicAMI patch from x's owned part
int ami_fill(const mi_matrix_s* m, const double* x)
{
    mi_addr_s* a = m->addr;
    double* ext = const_cast<double*>(x) + a->L.nCells;
    for (const mi_addr_s::AmiPatch* q : a->ami) {
        if (q->n == 0) continue;
        k_ami_fill<<<(q->n + 255) / 256, 256, 0, a->ctx->stream>>>(x, q->start.p, q->cellE.p, q->w.p, q->hasLow ? q->ownE.p : nullptr, patch_factor(m, q->patch),
                                                                    ext + q->extOff, q->n);
    }
    HIPCHK(hipGetLastError());
    return MI_OK;
}
// received processor-patch values of a transformed (processorCyclic) patch: processorGAMGInterfaceField.C:213,230
int scale_received(const mi_matrix_s* m, const double* x)
{
    if (m->patchFactor.empty()) return MI_OK;
    mi_addr_s* a = m->addr;
    for (int32_t p = 0; p < a->L.nPatches; ++p) {
        const double f = m->patchFactor[(size_t)p];
        const int n = a->L.patchOffset[(size_t)p + 1] - a->L.patchOffset[p];
        if (f == 1.0 || n == 0 || a->patchIsLocal[(size_t)p]) continue;
        k_scale_range<<<(n + 255) / 256, 256, 0, a->ctx->stream>>>(const_cast<double*>(x) + a->L.nCells + a->L.patchOffset[p], f, n);
    }
    HIPCHK(hipGetLastError());
    return MI_OK;
}

// Tile operator over all tiles.  With a communicator attached the neighbour values of the processor patches
// are exchanged here (init/updateMatrixInterfaces, lduMatrixUpdateMatrixInterfaces.C:30-276): pack, send/recv
// on the halo stream into x's ext region, interior tiles meanwhile, boundary tiles after the wait.  Without
// one the ext region holds whatever the caller placed there (mi_matrix_set_ext).
template <int OP>
int tile_op(mi_matrix_s* m, bool trans, const double* x, const double* b, const double* rD, double* y, double omega,
            double* dotPartial = nullptr, double* dotPartial2 = nullptr)
{
    constexpr bool readsNbr = (OP == OP_AMUL || OP == OP_RESIDUAL || OP == OP_H || OP == OP_JACOBI || OP == OP_PROLOGUE);
    mi_addr_s* a = m->addr;
    if (readsNbr && a->amiRemote && comm_remote(m)) {
        // a cyclicAMI patch whose partner lives on another rank interpolates from what its transport patch RECEIVES: whole
        // exchange first, then the interpolation, then all tiles in one launch (no interior / boundary overlap on such a matrix)
        if (peer_halo_ready(m)) { MICHK(peer_exchange_push(m, x)); MICHK(peer_exchange_pull(m, const_cast<double*>(x))); }
        else {
            if (m->sendBuf.n < (size_t)a->L.nExt) MICHK(m->sendBuf.alloc((size_t)a->L.nExt));
            MICHK(mi_halo_pack_engine(a, x, m->sendBuf.p));
            MICHK(comm_exchange_start(m, m->sendBuf.p, const_cast<double*>(x)));
            MICHK(comm_exchange_wait(m));
            MICHK(scale_received(m, x));
        }
        MICHK(ami_fill(m, x));
        return launch_tile<OP>(m, trans, x, b, rD, y, omega, 0, dotPartial, dotPartial2);
    }
    if (readsNbr && !a->ami.empty()) MICHK(ami_fill(m, x));
    if (!readsNbr || !comm_remote(m)) return launch_tile<OP>(m, trans, x, b, rD, y, omega, 0, dotPartial, dotPartial2);
    if (peer_halo_ready(m)) { // stores into the neighbours' windows instead of send/recv calls, no second stream
        // ONE launch: its first blocks push this rank's patch values, interior tiles follow, a boundary tile waits for the
        // neighbours' flags itself and reads the window (peer.inc: tile_kernel_win)
        if (peer_win_direct(m)) return launch_tile_win<OP>(m, trans, x, b, rD, y, omega, dotPartial, dotPartial2);
        MICHK(peer_exchange_push(m, x));
        MICHK(launch_tile<OP>(m, trans, x, b, rD, y, omega, 1, dotPartial, dotPartial2));
        MICHK(peer_exchange_pull(m, const_cast<double*>(x)));
        return launch_tile<OP>(m, trans, x, b, rD, y, omega, 2, dotPartial ? dotPartial + a->nInterior : nullptr, dotPartial2 ? dotPartial2 + a->nInterior : nullptr);
    }
    if (m->sendBuf.n < (size_t)a->L.nExt) MICHK(m->sendBuf.alloc((size_t)a->L.nExt));
    MICHK(mi_halo_pack_engine(a, x, m->sendBuf.p));
    MICHK(comm_exchange_start(m, m->sendBuf.p, const_cast<double*>(x)));
    MICHK(launch_tile<OP>(m, trans, x, b, rD, y, omega, 1, dotPartial, dotPartial2));
    MICHK(comm_exchange_wait(m));
    MICHK(scale_received(m, x));
    return launch_tile<OP>(m, trans, x, b, rD, y, omega, 2, dotPartial ? dotPartial + a->nInterior : nullptr, dotPartial2 ? dotPartial2 + a->nInterior : nullptr);
}

} // namespace

// ---------------------------------------------------------------------------
// layout helpers, halo
// ---------------------------------------------------------------------------
extern "C" int mi_vec_to_engine(mi_addr_t a, const double* x, double* xe)
{
    if (!a || !x || !xe) return fail(MI_ERR_ARG, "mi_vec_to_engine: bad argument");
    HIPCHK(hipSetDevice(a->ctx->device));
    k_gather_perm<<<RG, RB, 0, a->ctx->stream>>>(x, a->perm(), xe, a->L.nCells);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_vec_from_engine(mi_addr_t a, const double* xe, double* x)
{
    if (!a || !x || !xe) return fail(MI_ERR_ARG, "mi_vec_from_engine: bad argument");
    HIPCHK(hipSetDevice(a->ctx->device));
    k_scatter_perm<<<RG, RB, 0, a->ctx->stream>>>(xe, a->perm(), x, a->L.nCells);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_halo_pack_engine(mi_addr_t a, const double* xe, double* send)
{
    if (!a || !xe || (!send && a->L.nExt > 0)) return fail(MI_ERR_ARG, "mi_halo_pack_engine: bad argument");
    if (a->L.nExt == 0) return MI_OK;
    HIPCHK(hipSetDevice(a->ctx->device));
    k_halo_pack<<<(a->L.nExt + 255) / 256, 256, 0, a->ctx->stream>>>(xe, a->patchFaceCellsE.p, send, a->L.nExt);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------------------
// SpMV family
// ---------------------------------------------------------------------------
extern "C" int mi_amul_engine(mi_matrix_t m, const double* psi_e, double* Apsi_e, int which)
{
    if (!m || !psi_e || !Apsi_e) return fail(MI_ERR_ARG, "mi_amul_engine: bad argument");
    HIPCHK(hipSetDevice(m->addr->ctx->device));
    if (which == 0) return tile_op<OP_AMUL>(m, false, psi_e, nullptr, nullptr, Apsi_e, 0.0); // exchanges the halo when attached
    return launch_tile<OP_AMUL>(m, false, psi_e, nullptr, nullptr, Apsi_e, 0.0, which);
}
extern "C" int mi_tmul_engine(mi_matrix_t m, const double* psi_e, double* Tpsi_e, int which)
{
    if (!m || !psi_e || !Tpsi_e) return fail(MI_ERR_ARG, "mi_tmul_engine: bad argument");
    HIPCHK(hipSetDevice(m->addr->ctx->device));
    if (which == 0) return tile_op<OP_AMUL>(m, true, psi_e, nullptr, nullptr, Tpsi_e, 0.0);
    return launch_tile<OP_AMUL>(m, true, psi_e, nullptr, nullptr, Tpsi_e, 0.0, which);
}

extern "C" int mi_matrix_set_ext(mi_matrix_t m, const double* ext)
{
    if (!m) return fail(MI_ERR_ARG, "mi_matrix_set_ext: bad argument");
    const int32_t nExt = m->addr->L.nExt;
    if (nExt == 0) return MI_OK;
    if (!ext) return fail(MI_ERR_ARG, "mi_matrix_set_ext: ext is NULL");
    HIPCHK(hipSetDevice(m->addr->ctx->device));
    double* v0;
    MICHK(m->vec(0, &v0));
    HIPCHK(hipMemcpyAsync(v0 + m->addr->L.nCells, ext, sizeof(double) * (size_t)nExt, hipMemcpyDeviceToDevice, m->addr->ctx->stream));
    return MI_OK;
}

namespace {
// caller-order wrapper: x -> engine (work 0, keeps the ext tail), op -> work 1, -> caller
template <int OP>
int caller_op(mi_matrix_s* m, bool trans, const double* x, const double* b, double* y)
{
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    hipStream_t s = a->ctx->stream;
    double *v0, *v1, *v2 = nullptr;
    if (a->identity) {
        // ordered addressing: no permutation passes.  The input only has to be copied when the operator reads an ext
        // region (coupled patches whose neighbour values live behind the n_cells owned values of an engine vector).
        const double* xin = x;
        if (x && a->L.nExt > 0) { MICHK(m->vec(0, &v0)); HIPCHK(hipMemcpyAsync(v0, x, sizeof(double) * (size_t)a->L.nCells, hipMemcpyDeviceToDevice, s)); xin = v0; }
        return tile_op<OP>(m, trans, xin, b, nullptr, y, 0.0);
    }
    constexpr bool fusable = (OP == OP_AMUL || OP == OP_RESIDUAL || OP == OP_H || OP == OP_SUMA || OP == OP_H1);
    if (fusable && a->ctx->fusePerm && !comm_remote(m) && a->ami.empty() && !a->compact) {
        // the permutation folded into the tile kernel: it gathers x[e2c] while staging and stores y[e2c] -- no separate passes.
        // (The ext tail, if the mesh has coupled patches whose values the caller placed with mi_matrix_set_ext, lives in work 0.)
        MICHK(m->vec(0, &v0));
        if (a->haloSrc.n != a->haloCell.n) {
            Table<int32_t> hc(a->haloCell.n), src(a->haloCell.n);
            HIPCHK(hipMemcpy(hc.data(), a->haloCell.p, sizeof(int32_t) * hc.size(), hipMemcpyDeviceToHost));
            for (size_t h = 0; h < hc.size(); ++h) src[h] = hc[h] < a->L.nCells ? a->L.e2c[(size_t)hc[h]] : -1 - (hc[h] - a->L.nCells);
            MICHK(a->haloSrc.upload(src, s));
            HIPCHK(hipStreamSynchronize(s));
        }
        m->callerX = x; m->callerB = b; m->callerY = y;
        const int rc = launch_tile<OP>(m, trans, v0, nullptr, nullptr, nullptr, 0.0, 0);
        m->callerX = nullptr; m->callerB = nullptr; m->callerY = nullptr;
        return rc;
    }
    MICHK(m->vec(0, &v0)); MICHK(m->vec(1, &v1));
    if (x) k_gather_perm<<<RG, RB, 0, s>>>(x, a->perm(), v0, a->L.nCells);
    if (b) { MICHK(m->vec(2, &v2)); k_gather_perm<<<RG, RB, 0, s>>>(b, a->perm(), v2, a->L.nCells); }
    MICHK(tile_op<OP>(m, trans, v0, v2, nullptr, v1, 0.0));
    k_scatter_perm<<<RG, RB, 0, s>>>(v1, a->perm(), y, a->L.nCells);
    HIPCHK(hipGetLastError());
    return MI_OK;
}
} // namespace

extern "C" int mi_amul(mi_matrix_t m, const double* psi, double* Apsi)
{
    if (!m || !psi || !Apsi) return fail(MI_ERR_ARG, "mi_amul: bad argument");
    return caller_op<OP_AMUL>(m, false, psi, nullptr, Apsi);
}
extern "C" int mi_tmul(mi_matrix_t m, const double* psi, double* Tpsi)
{
    if (!m || !psi || !Tpsi) return fail(MI_ERR_ARG, "mi_tmul: bad argument");
    return caller_op<OP_AMUL>(m, true, psi, nullptr, Tpsi);
}
extern "C" int mi_sumA(mi_matrix_t m, double* sumA)
{
    if (!m || !sumA) return fail(MI_ERR_ARG, "mi_sumA: bad argument");
    return caller_op<OP_SUMA>(m, false, nullptr, nullptr, sumA);
}
extern "C" int mi_residual(mi_matrix_t m, const double* psi, const double* source, double* rA)
{
    if (!m || !psi || !source || !rA) return fail(MI_ERR_ARG, "mi_residual: bad argument");
    return caller_op<OP_RESIDUAL>(m, false, psi, source, rA);
}
extern "C" int mi_H(mi_matrix_t m, const double* psi, double* H)
{
    if (!m || !psi || !H) return fail(MI_ERR_ARG, "mi_H: bad argument");
    return caller_op<OP_H>(m, false, psi, nullptr, H);
}
extern "C" int mi_H1(mi_matrix_t m, double* H1)
{
    if (!m || !H1) return fail(MI_ERR_ARG, "mi_H1: bad argument");
    return caller_op<OP_H1>(m, false, nullptr, nullptr, H1);
}

extern "C" int mi_faceH(mi_matrix_t m, const double* psi, double* faceH)
{
    if (!m || !psi || !faceH) return fail(MI_ERR_ARG, "mi_faceH: bad argument");
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    hipStream_t s = a->ctx->stream;
    if (a->lowerAddr.n != (size_t)a->L.nFaces) { MICHK(a->lowerAddr.upload(a->lowerHost, s)); MICHK(a->upperAddr.upload(a->upperHost, s)); }
    const int nF = a->L.nFaces;
    if (nF == 0) return MI_OK;
    k_faceH<<<2048, 256, 0, s>>>(psi, a->lowerAddr.p, a->upperAddr.p, a->faceSlot.p, m->upE.p, m->asym ? m->lowE.p : m->upE.p, faceH, nF);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

namespace {
// diagonal preconditioner on the caller's arrays of a permuting addressing: wA[c] = rD[c2e(c)] * rA[c], written as one pass
// over the engine cells (rD is stored in engine order) instead of gather + multiply + scatter
__global__ __launch_bounds__(256) void k_mul_perm(double* __restrict__ w, const double* __restrict__ rD, const double* __restrict__ r,
                                                  const int32_t* __restrict__ e2c, int n)
{
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) { const int c = e2c[e]; w[c] = rD[e] * r[c]; }
}
} // namespace

extern "C" int mi_precondition(mi_matrix_t m, int kind, int transpose, const double* rA, double* wA)
{
    if (!m || !rA || !wA) return fail(MI_ERR_ARG, "mi_precondition: bad argument");
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    hipStream_t s = a->ctx->stream;
    double *v0, *v1;
    if (a->identity && wA != rA) { // ordered addressing: straight on the caller's arrays (the preconditioners read no ext region)
        if (kind == MI_PRECOND_NONE) { HIPCHK(hipMemcpyAsync(wA, rA, sizeof(double) * (size_t)a->L.nCells, hipMemcpyDeviceToDevice, s)); return MI_OK; }
        if (kind != MI_PRECOND_DIAGONAL && kind != MI_PRECOND_AINV) return fail(MI_ERR_ARG, "unknown preconditioner kind");
        MICHK(ensure_rD(m));
        if (kind == MI_PRECOND_DIAGONAL && aligned16(rA) && aligned16(wA)) { k_mul<<<RG, RB, 0, s>>>(wA, m->rD.p, rA, a->L.nCells); HIPCHK(hipGetLastError()); return MI_OK; }
        // (the AINV tile pass stages x[haloCell] for every halo entry, ext entries included -- their values are never used, but
        //  the caller's rA ends at n_cells: with an ext region the input goes through work vector 0 below, as caller_op does)
        if (kind == MI_PRECOND_AINV && a->L.nExt == 0) return launch_tile<OP_AINV>(m, transpose != 0, rA, nullptr, m->rD.p, wA, 0.0, 0);
    }
    if (!a->identity && wA != rA && a->ctx->fusePerm) { // no permutation passes for the two preconditioners that are pointwise
        if (kind == MI_PRECOND_NONE) { HIPCHK(hipMemcpyAsync(wA, rA, sizeof(double) * (size_t)a->L.nCells, hipMemcpyDeviceToDevice, s)); return MI_OK; }
        if (kind == MI_PRECOND_DIAGONAL) {
            MICHK(ensure_rD(m));
            k_mul_perm<<<4096, 256, 0, s>>>(wA, m->rD.p, rA, a->e2c.p, a->L.nCells);
            HIPCHK(hipGetLastError());
            return MI_OK;
        }
    }
    MICHK(m->vec(0, &v0)); MICHK(m->vec(1, &v1));
    k_gather_perm<<<RG, RB, 0, s>>>(rA, a->perm(), v0, a->L.nCells);
    if (kind == MI_PRECOND_NONE) {
        HIPCHK(hipMemcpyAsync(v1, v0, sizeof(double) * (size_t)a->L.nCells, hipMemcpyDeviceToDevice, s));
    } else if (kind == MI_PRECOND_DIAGONAL) {
        MICHK(ensure_rD(m));
        k_mul<<<RG, RB, 0, s>>>(v1, m->rD.p, v0, a->L.nCells);
    } else if (kind == MI_PRECOND_AINV) {
        MICHK(ensure_rD(m));
        MICHK(launch_tile<OP_AINV>(m, transpose != 0, v0, nullptr, m->rD.p, v1, 0.0, 0));
    } else return fail(MI_ERR_ARG, "unknown preconditioner kind");
    k_scatter_perm<<<RG, RB, 0, s>>>(v1, a->perm(), wA, a->L.nCells);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_jacobi_smooth(mi_matrix_t m, double omega, double* psi, const double* source, int32_t n_sweeps)
{
    if (!m || !psi || !source || n_sweeps < 0) return fail(MI_ERR_ARG, "mi_jacobi_smooth: bad argument");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    hipStream_t s = a->ctx->stream;
    double *v0, *v1, *v2;
    MICHK(m->vec(0, &v0)); MICHK(m->vec(1, &v1)); MICHK(m->vec(2, &v2));
    if (a->identity && a->L.nExt == 0) { // ordered addressing, no ext region: ping-pong between the caller's psi and one work vector
        double *cur = psi, *nxt = v1;
        for (int sw = 0; sw < n_sweeps; ++sw) { MICHK(tile_op<OP_JACOBI>(m, false, cur, source, nullptr, nxt, omega)); double* t = cur; cur = nxt; nxt = t; }
        if (cur != psi) HIPCHK(hipMemcpyAsync(psi, cur, sizeof(double) * (size_t)a->L.nCells, hipMemcpyDeviceToDevice, s));
        return MI_OK;
    }
    k_gather_perm<<<RG, RB, 0, s>>>(psi, a->perm(), v0, a->L.nCells);
    k_gather_perm<<<RG, RB, 0, s>>>(source, a->perm(), v2, a->L.nCells);
    double *cur = v0, *nxt = v1;
    for (int sw = 0; sw < n_sweeps; ++sw) {
        // ping-pong instead of the reference's `psi = Apsi` copy (JacobiSmoother.C:146)
        if (a->L.nExt > 0 && sw > 0 && !comm_remote(m))
            HIPCHK(hipMemcpyAsync(cur + a->L.nCells, nxt + a->L.nCells, sizeof(double) * (size_t)a->L.nExt, hipMemcpyDeviceToDevice, s));
        MICHK(tile_op<OP_JACOBI>(m, false, cur, v2, nullptr, nxt, omega));
        double* t = cur; cur = nxt; nxt = t;
    }
    k_scatter_perm<<<RG, RB, 0, s>>>(cur, a->perm(), psi, a->L.nCells);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------------------
// reductions (host result)
// ---------------------------------------------------------------------------
namespace {
template <int KIND>
int reduce_host(mi_ctx_s* c, const double* a, const double* b, int64_t n, double* out)
{
    if (!c || !a || !out || n < 0) return fail(MI_ERR_ARG, "reduction: bad argument");
    if (c->session) return fail(MI_ERR_STATE, "reduction: a PCG session (mi_pcg_begin) is active on this context and owns its reduction scratch; call mi_pcg_end first");
    if (!aligned16(a) || (b && !aligned16(b))) return fail(MI_ERR_ARG, "reduction inputs must be 16-byte aligned");
    HIPCHK(hipSetDevice(c->device));
    k_reduce<KIND><<<RG, RB, 0, c->stream>>>(a, b, n, c->partial.p);
    k_reduce_final<<<1, RB, 0, c->stream>>>(c->partial.p, c->scalars.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->hostScal, c->scalars.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = c->hostScal[0];
    return MI_OK;
}
} // namespace

extern "C" int mi_sum(mi_ctx_t c, const double* a, int64_t n, double* out) { return reduce_host<RED_SUM>(c, a, nullptr, n, out); }
extern "C" int mi_sum_prod(mi_ctx_t c, const double* a, const double* b, int64_t n, double* out)
{
    if (!b) return fail(MI_ERR_ARG, "mi_sum_prod: b is NULL");
    return reduce_host<RED_PROD>(c, a, b, n, out);
}
extern "C" int mi_sum_mag(mi_ctx_t c, const double* a, int64_t n, double* out) { return reduce_host<RED_MAG>(c, a, nullptr, n, out); }

// ---------------------------------------------------------------------------
// solvers
// ---------------------------------------------------------------------------
namespace {
int reduce_sync_fwd(mi_matrix_s* m, const double* a, double* out);
}

// ---- engine-order primitives for a caller that keeps its solver loop (level-1 integration) but lets its work vectors live
//      in engine order for the duration of a solve (mi_vec_to_engine once, mi_vec_from_engine once): no permutation passes.
//      Vectors are n_cells + n_ext long.
namespace { int precond_engine(mi_matrix_s* m, int kind, bool transpose, const double* r, double* w); }
extern "C" int mi_precondition_engine(mi_matrix_t m, int kind, int transpose, const double* rA_e, double* wA_e)
{
    if (!m || !rA_e || !wA_e) return fail(MI_ERR_ARG, "mi_precondition_engine: bad argument");
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound");
    HIPCHK(hipSetDevice(m->addr->ctx->device));
    return precond_engine(m, kind, transpose != 0, rA_e, wA_e);
}
extern "C" int mi_residual_engine(mi_matrix_t m, const double* psi_e, const double* source_e, double* rA_e)
{
    if (!m || !psi_e || !source_e || !rA_e) return fail(MI_ERR_ARG, "mi_residual_engine: bad argument");
    HIPCHK(hipSetDevice(m->addr->ctx->device));
    return tile_op<OP_RESIDUAL>(m, false, psi_e, source_e, nullptr, rA_e, 0.0);
}
extern "C" int mi_jacobi_smooth_engine(mi_matrix_t m, double omega, double* psi_e, const double* source_e, int32_t n_sweeps)
{
    if (!m || !psi_e || !source_e || n_sweeps < 0) return fail(MI_ERR_ARG, "mi_jacobi_smooth_engine: bad argument");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    hipStream_t s = a->ctx->stream;
    double* scratch;
    MICHK(m->vec(1, &scratch));
    double *cur = psi_e, *nxt = scratch;
    for (int sw = 0; sw < n_sweeps; ++sw) {
        if (a->L.nExt > 0 && sw > 0 && !comm_remote(m))
            HIPCHK(hipMemcpyAsync(cur + a->L.nCells, nxt + a->L.nCells, sizeof(double) * (size_t)a->L.nExt, hipMemcpyDeviceToDevice, s));
        MICHK(tile_op<OP_JACOBI>(m, false, cur, source_e, nullptr, nxt, omega));
        double* t = cur; cur = nxt; nxt = t;
    }
    if (cur != psi_e) HIPCHK(hipMemcpyAsync(psi_e, cur, sizeof(double) * (size_t)a->L.nCells, hipMemcpyDeviceToDevice, s));
    return MI_OK;
}

// lduMatrix::solver::normFactor (lduMatrixSolver.C:182-236) for caller-order psi/source/Apsi. The
// vectors are gathered into engine order first, so the value is bit-identical to the normFactor
// the solvers' own prologue computes (and global when a communicator is attached).
namespace {
int norm_factor_engine(mi_matrix_s* m, const double* psi_e, const double* src_e, const double* Apsi_e, double* sumA_scratch, double* out)
{
    mi_addr_s* a = m->addr;
    mi_ctx_s* c = a->ctx;
    hipStream_t s = c->stream;
    const int64_t n = a->L.nCells;
    MICHK(launch_tile<OP_SUMA>(m, false, nullptr, nullptr, nullptr, sumA_scratch, 0.0, 0));
    double sumPsi = 0;
    MICHK(reduce_sync_fwd(m, psi_e, &sumPsi));
    const double avg = sumPsi / (double)(comm_attached(m) ? comm_n_global(m) : n);
    k_normfactor<<<RG, RB, 0, s>>>(Apsi_e, src_e, sumA_scratch, avg, n, c->partial.p);
    k_reduce_final<<<1, RB, 0, s>>>(c->partial.p, c->scalars.p);
    HIPCHK(hipGetLastError());
    if (comm_attached(m)) MICHK(comm_allreduce(m, c->scalars.p, 1));
    HIPCHK(hipMemcpyAsync(c->hostScal, c->scalars.p, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *out = c->hostScal[0] + SP_SMALL; // + matrix.small_ (lduMatrixSolver.C:228)
    return MI_OK;
}
} // namespace
extern "C" int mi_norm_factor(mi_matrix_t m, const double* psi, const double* source, const double* Apsi, double* out)
{
    if (!m || !psi || !source || !Apsi || !out) return fail(MI_ERR_ARG, "mi_norm_factor: bad argument");
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound");
    if (m->addr->ctx->session) return fail(MI_ERR_STATE, "mi_norm_factor: a PCG session (mi_pcg_begin) is active on this context; call mi_pcg_end first");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    hipStream_t s = a->ctx->stream;
    const int64_t n = a->L.nCells;
    double *v0, *v1, *v2, *v3;
    MICHK(m->vec(0, &v0)); MICHK(m->vec(1, &v1)); MICHK(m->vec(2, &v2)); MICHK(m->vec(3, &v3));
    k_gather_perm<<<RG, RB, 0, s>>>(Apsi, a->perm(), v0, n);
    k_gather_perm<<<RG, RB, 0, s>>>(source, a->perm(), v2, n);
    k_gather_perm<<<RG, RB, 0, s>>>(psi, a->perm(), v3, n);
    return norm_factor_engine(m, v3, v2, v0, v1, out);
}
extern "C" int mi_norm_factor_engine(mi_matrix_t m, const double* psi_e, const double* source_e, const double* Apsi_e, double* out)
{
    if (!m || !psi_e || !source_e || !Apsi_e || !out) return fail(MI_ERR_ARG, "mi_norm_factor_engine: bad argument");
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound");
    if (m->addr->ctx->session) return fail(MI_ERR_STATE, "mi_norm_factor_engine: a PCG session (mi_pcg_begin) is active on this context; call mi_pcg_end first");
    HIPCHK(hipSetDevice(m->addr->ctx->device));
    double* v1;
    MICHK(m->vec(1, &v1));
    return norm_factor_engine(m, psi_e, source_e, Apsi_e, v1, out);
}

namespace {

// engine-order preconditioner application
int precond_engine(mi_matrix_s* m, int kind, bool transpose, const double* r, double* w)
{
    mi_addr_s* a = m->addr;
    hipStream_t s = a->ctx->stream;
    const int64_t n = a->L.nCells;
    if (kind == MI_PRECOND_NONE) { HIPCHK(hipMemcpyAsync(w, r, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s)); return MI_OK; }
    MICHK(ensure_rD(m));
    if (kind == MI_PRECOND_DIAGONAL) { k_mul<<<RG, RB, 0, s>>>(w, m->rD.p, r, n); HIPCHK(hipGetLastError()); return MI_OK; }
    if (kind == MI_PRECOND_AINV) return launch_tile<OP_AINV>(m, transpose, r, nullptr, m->rD.p, w, 0.0, 0);
    return fail(MI_ERR_ARG, "unknown preconditioner kind");
}

// host-visible scalar reduction on engine vectors
template <int KIND>
int reduce_sync(mi_matrix_s* m, const double* a, const double* b, double* out)
{
    mi_ctx_s* c = m->addr->ctx;
    k_reduce<KIND><<<RG, RB, 0, c->stream>>>(a, b, (int64_t)m->addr->L.nCells, c->partial.p);
    k_reduce_final<<<1, RB, 0, c->stream>>>(c->partial.p, c->scalars.p);
    HIPCHK(hipGetLastError());
    if (comm_attached(m)) MICHK(comm_allreduce(m, c->scalars.p, 1)); // Foam::reduce(..., sumOp<scalar>())
    HIPCHK(hipMemcpyAsync(c->hostScal, c->scalars.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (comm_attached(m)) MICHK(peer_check(m));   // a window wait that ran out of polls voids the sum
    *out = c->hostScal[0];
    return MI_OK;
}

int reduce_sync_fwd(mi_matrix_s* m, const double* a, double* out) { return reduce_sync<RED_SUM>(m, a, nullptr, out); }

// common start of every solver (PCG.C:91-121 and siblings): given psi_e, src_e:
//   wA = A psi ; rA = src - wA ; normFactor ; initial residual ; convergence test.
// Leaves the result in the device PcgState; tmp is scratch (the reference passes pA).
int solve_prologue(mi_matrix_s* m, const mi_solver_controls* ctl, const double* psi_e, const double* src_e,
                   double* wA, double* rA, double* tmp, int histLen)
{
    mi_addr_s* a = m->addr;
    mi_ctx_s* c = a->ctx;
    hipStream_t s = c->stream;
    const int64_t n = a->L.nCells;
    if (m->hist.n < (size_t)(histLen > 0 ? histLen : 1)) MICHK(m->hist.alloc((size_t)(histLen > 0 ? histLen : 1)));
    m->histLen = histLen;
    PcgState h;
    memset(&h, 0, sizeof(h));
    h.tolerance = ctl->tolerance; h.relTol = ctl->relTol; h.maxIter = ctl->maxIter; h.minIter = ctl->minIter;
    *c->hostState = h;
    HIPCHK(hipMemcpyAsync(c->state.p, c->hostState, sizeof(PcgState), hipMemcpyHostToDevice, s));
    (void)tmp;
    if (c->fusePrologue) {
        // ONE pass over the coefficients: wA = A psi, rA = source - wA and -- when the coefficients were re-bound since the last solve, as in
        // every equation of a time step -- sumA (OP_PROLOGUE: the Amul's and the sumA pass's chains side by side, same bits); then the
        // sum of psi, and normFactor + sum|rA| in one vector pass with gAverage(psi) formed on the device (no host read)
        const bool needSumA = !m->sumAValid;
        if (needSumA && m->sumAE.n != (size_t)n) MICHK(m->sumAE.alloc((size_t)n));
        m->proY2 = rA; m->proY3 = needSumA ? m->sumAE.p : nullptr;
        const int rc = tile_op<OP_PROLOGUE>(m, false, psi_e, src_e, nullptr, wA, 0.0);
        m->proY2 = nullptr; m->proY3 = nullptr;
        MICHK(rc);
        m->sumAValid = true;
        double* P = c->partial.p;          // [0, RG): normFactor, [RG, 2 RG): sum|rA|, [2 RG, 3 RG): sum psi
        k_reduce<RED_SUM><<<RG, RB, 0, s>>>(psi_e, nullptr, n, P + 2 * RG);
        if (comm_attached(m)) { // gSum(psi), then the two sums are global (lduMatrixSolver.C:182-236, gSumMag)
            k_reduce_final<<<1, RB, 0, s>>>(P + 2 * RG, c->scalars.p);
            MICHK(comm_allreduce(m, c->scalars.p, 1));
            k_normfactor_mag<true><<<RG, RB, 0, s>>>(wA, src_e, m->sumAE.p, c->scalars.p, (double)comm_n_global(m), n, P, P + RG);
            k_reduce_final2<<<2, RB, 0, s>>>(P, c->scalars.p + 1, P + RG, c->scalars.p + 2);
            MICHK(comm_allreduce(m, c->scalars.p + 1, 2));
            k_solve_init<true><<<1, RB, 0, s>>>(c->state.p, c->scalars.p + 1, c->scalars.p + 2, m->hist.p, histLen);
        } else {
            k_normfactor_mag<false><<<RG, RB, 0, s>>>(wA, src_e, m->sumAE.p, P + 2 * RG, (double)n, n, P, P + RG);
            k_solve_init<false><<<1, RB, 0, s>>>(c->state.p, P, P + RG, m->hist.p, histLen);
        }
        HIPCHK(hipGetLastError());
        return MI_OK;
    }
    MICHK(tile_op<OP_AMUL>(m, false, psi_e, nullptr, nullptr, wA, 0.0));
    k_sub<<<RG, RB, 0, s>>>(rA, src_e, wA, n);
    if (!m->sumAValid) { // sumA depends on the coefficients only: once per binding, not once per solve
        if (m->sumAE.n != (size_t)n) MICHK(m->sumAE.alloc((size_t)n));
        MICHK(launch_tile<OP_SUMA>(m, false, nullptr, nullptr, nullptr, m->sumAE.p, 0.0, 0));
        m->sumAValid = true;
    }
    // gAverage(psi) (gpuFieldCommonFunctions.C:611-634): needs the host for the division by N
    double sumPsi = 0;
    MICHK(reduce_sync<RED_SUM>(m, psi_e, nullptr, &sumPsi));
    const double avg = sumPsi / (double)(comm_attached(m) ? comm_n_global(m) : n);
    k_normfactor<<<RG, RB, 0, s>>>(wA, src_e, m->sumAE.p, avg, n, c->partial.p);
    k_reduce<RED_MAG><<<RG, RB, 0, s>>>(rA, nullptr, n, c->partial.p + RG);
    if (comm_attached(m)) { // the two sums are global (lduMatrixSolver.C:182-236, gSumMag)
        k_reduce_final2<<<2, RB, 0, s>>>(c->partial.p, c->scalars.p + 1, c->partial.p + RG, c->scalars.p + 2);
        MICHK(comm_allreduce(m, c->scalars.p + 1, 2));
        k_solve_init<true><<<1, RB, 0, s>>>(c->state.p, c->scalars.p + 1, c->scalars.p + 2, m->hist.p, histLen);
    } else
    k_solve_init<false><<<1, RB, 0, s>>>(c->state.p, c->partial.p, c->partial.p + RG, m->hist.p, histLen);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

int fetch_state(mi_ctx_s* c)
{
    HIPCHK(hipMemcpyAsync(c->hostState, c->state.p, sizeof(PcgState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->hostState->fault) {
        // the barrier words of the matrix that ran are out of step now: every matrix of this context re-zeroes its own before
        // its next persistent launch (persist_enqueue); the next solve's prologue clears PcgState::fault itself
        c->faultEpoch++;
        return fail(MI_ERR_DEVICE, "persistent PCG kernel: a workgroup never reached a grid barrier (ran out of polls); the results of this solve are not valid");
    }
    return MI_OK;
}

void fill_perf(const PcgState& h, mi_solver_perf* p)
{
    p->initialResidual = h.initialResidual; p->finalResidual = h.finalResidual; p->normFactor = h.normFactor;
    p->nIterations = h.nIterations; p->converged = h.converged; p->singular = h.singular; p->reserved = 0;
}

int copy_hist(mi_matrix_s* m, double* hist_host, int len, int nIter)
{
    if (!hist_host || len <= 0) return MI_OK;
    int cnt = nIter + 1; if (cnt > len) cnt = len; if (cnt > m->histLen) cnt = m->histLen;
    if (cnt > 0) {
        HIPCHK(hipMemcpyAsync(hist_host, m->hist.p, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost, m->addr->ctx->stream));
        HIPCHK(hipStreamSynchronize(m->addr->ctx->stream));
    }
    return MI_OK;
}

// Device-resident solver loops on a communicator-attached matrix: a local sum (RG block partials) becomes the GLOBAL sum
// where the reference calls Foam::reduce (allReduceTemplates.C:195-208) -- final reduction, all-reduce on the engine's stream,
// spread back as {sum, 0, ...} -- and the consumers read it exactly as they read local partials; the host still only polls
// `done` once per batch.  Two sums that are due at the same point travel in one all-reduce.
int globalize(mi_matrix_s* m, double* PA, double* PB = nullptr)
{
    if (!comm_attached(m)) return MI_OK;
    if (peer_reduce_ready(m)) { double* P[2] = {PA, PB}; return peer_globalize(m, PB ? 2 : 1, P); }   // final reduction, window all-reduce and spread in one launch
    mi_ctx_s* c = m->addr->ctx;
    hipStream_t s = c->stream;
    double* sc = c->scalars.p + 10;
    if (PB) k_reduce_final2<<<2, RB, 0, s>>>(PA, sc, PB, sc + 1);
    else k_reduce_final<<<1, RB, 0, s>>>(PA, sc);
    MICHK(comm_allreduce(m, sc, PB ? 2 : 1));
    k_spread_partials<<<PB ? 2 : 1, RB, 0, s>>>(sc, PA, PB);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

bool pcg_fused_rp_usable(const mi_matrix_s* m, int precond);      // pcg_fused.inc: residual update of iteration it + direction update of it + 1 in one launch
int pcg_fused_rp_launch(mi_matrix_s* m, int it, int precond);
bool pcg_persist_usable(const mi_matrix_s* m, int precond);       // persist.inc: the iteration as one persistent cooperative kernel
int pcg_persist_enqueue(mi_matrix_s* m, int n_iters, int precond);
// enqueue PCG iteration bodies it0 .. it0+count-1 (no host sync).
// diagonal / none: 5 launches per iteration -- update_p (precondition fused), Amul (+ fused
// gSumProd partials), fold, update_psi_r (+ next iteration's wArA partials), final.
// AINV: the preconditioner is itself a tile pass, so wA is materialised.
int pcg_enqueue(mi_matrix_s* m, int it0, int count, int precond, int evStride) // evStride > 0: HIP events around every evStride-th Amul
{
    mi_addr_s* a = m->addr;
    mi_ctx_s* c = a->ctx;
    hipStream_t s = c->stream;
    const int64_t n = a->L.nCells;
    double *psi, *src, *pA, *wA, *rA;
    MICHK(m->vec(3, &psi)); MICHK(m->vec(4, &src)); MICHK(m->vec(5, &pA)); MICHK(m->vec(6, &wA)); MICHK(m->vec(7, &rA));
    double* P1 = c->partial.p; double* P2 = c->partial.p + RG; double* P3 = c->partial.p + 2 * RG;
    if (precond != MI_PRECOND_NONE) MICHK(ensure_rD(m));
    if (m->tilePartial.n < (size_t)a->L.nTiles) MICHK(m->tilePartial.alloc((size_t)a->L.nTiles));
    const bool fuse = c->fuseFinal && precond != MI_PRECOND_AINV;
    const int defer = (c->deferPsi && !fuse) ? 1 : 0; // (the fused convergence test has no per-iteration k_pcg_final to record the added psi term)
    double* psiD = defer ? psi : nullptr;
    struct Gate { mi_matrix_s* m; explicit Gate(mi_matrix_s* mm) : m(mm) { m->gateDone = true; } ~Gate() { m->gateDone = false; } } gate(m);
    // round 6: k_pcg_update_psi_r(it) + k_pcg_final(it) + k_pcg_update_p(it + 1) as one launch that keeps rD o rA on the chip
    // (pcg_fused.inc).  Not under graph capture (it0 < 0: the launch carries the iteration number and its barrier generation).
    bool fusedRP = it0 >= 0 && pcg_fused_rp_usable(m, precond);
    if (it0 < 0) m->pcgPReady = false;
    for (int k = 0; k < count; ++k) {
        const int it = it0 < 0 ? -1 : it0 + k; // it0 < 0: the iteration counter lives on the device (graph replay)
        if (m->pcgPReady) {
            // pA of this iteration came out of the previous iteration's fused launch
        } else
        if (precond == MI_PRECOND_AINV) {
            // AINV apply with sum wA.rA fused into the tile pass (per-tile partials folded into P1): no separate reduction pass
            MICHK(launch_tile<OP_AINV>(m, false, rA, nullptr, m->rD.p, wA, 0.0, 0, m->tilePartial.p));
            k_fold_partials<<<1, 1024, 0, s>>>(m->tilePartial.p, a->L.nTiles, P1);
            MICHK(globalize(m, P1));
            k_pcg_update_p<0><<<RG, RB, 0, s>>>(c->state.p, it, P1, wA, nullptr, nullptr, pA, n, nullptr, nullptr, 0, psiD); // AINV path keeps k_pcg_final (P1 is rewritten before it)
        } else if (precond == MI_PRECOND_DIAGONAL) {
            k_pcg_update_p<1><<<RG, RB, 0, s>>>(c->state.p, it, P1, nullptr, m->rD.p, rA, pA, n, fuse ? P3 : nullptr, m->hist.p, m->histLen, psiD);
        } else {
            k_pcg_update_p<2><<<RG, RB, 0, s>>>(c->state.p, it, P1, nullptr, nullptr, rA, pA, n, fuse ? P3 : nullptr, m->hist.p, m->histLen, psiD);
        }
        const bool rec = evStride > 0 && (k % evStride) == 0;
        const size_t ev = rec ? (size_t)2 * (size_t)(k / evStride) : 0;
        if (rec) { // the event pair rides on the Amul launch itself: it brackets the kernel, not the launch gap before it
            while (m->evPool.size() < ev + 2) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); m->evPool.push_back(e); }
            if (c->attachEvents) { m->kevStart = m->evPool[ev]; m->kevStop = m->evPool[ev + 1]; }
            else HIPCHK(hipEventRecord(m->evPool[ev], s));
        }
        const int rcA = tile_op<OP_AMUL>(m, false, pA, nullptr, nullptr, wA, 0.0, m->tilePartial.p);   // exchanges the halo when attached
        m->kevStart = nullptr; m->kevStop = nullptr;
        MICHK(rcA);
        if (rec && !c->attachEvents) HIPCHK(hipEventRecord(m->evPool[ev + 1], s));
        k_fold_partials<<<1, 1024, 0, s>>>(m->tilePartial.p, a->L.nTiles, P2);
        MICHK(globalize(m, P2));
        m->pcgPReady = false;
        if (fusedRP) {
            const int rcF = pcg_fused_rp_launch(m, it, precond);
            if (rcF == MI_OK) { m->pcgPReady = true; continue; }
            if (rcF != MI_ERR_UNSUPPORTED) return rcF;
            fusedRP = false;   // the device does not hold the grid: the separate kernels, here and from now on
        }
        if (precond == MI_PRECOND_AINV)
            k_pcg_update_psi_r<0><<<RG, RB, 0, s>>>(c->state.p, it, P2, pA, wA, nullptr, psi, rA, n, P3, P1, defer);
        else if (precond == MI_PRECOND_DIAGONAL)
            k_pcg_update_psi_r<1><<<RG, RB, 0, s>>>(c->state.p, it, P2, pA, wA, m->rD.p, psi, rA, n, P3, P1, defer);
        else
            k_pcg_update_psi_r<2><<<RG, RB, 0, s>>>(c->state.p, it, P2, pA, wA, nullptr, psi, rA, n, P3, P1, defer);
        MICHK(globalize(m, P3, precond == MI_PRECOND_AINV ? nullptr : P1));   // sum|rA| (+ the next iteration's wA.rA from the same pass)
        // diagonal / none: the convergence test of this iteration is fused into the next k_pcg_update_p
        if (!fuse) k_pcg_final<false><<<1, RB, 0, s>>>(c->state.p, it, P3, m->hist.p, m->histLen);
    }
    // test of the last enqueued iteration (idempotent: the next batch's first kernel repeats it)
    if (count > 0 && fuse) k_pcg_final<false><<<1, RB, 0, s>>>(c->state.p, it0 < 0 ? -1 : it0 + count - 1, P3, m->hist.p, m->histLen);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

// end of a pcg_enqueue-driven solve: the psi term the last iteration still owes (deferred psi update)
int pcg_flush(mi_matrix_s* m)
{
    mi_ctx_s* c = m->addr->ctx;
    if (!c->deferPsi) return MI_OK;
    double *psi, *pA;
    MICHK(m->vec(3, &psi)); MICHK(m->vec(5, &pA));
    k_pcg_flush_psi<<<RG, RB, 0, c->stream>>>(c->state.p, pA, psi, m->addr->L.nCells);
    k_pcg_flush_mark<<<1, 1, 0, c->stream>>>(c->state.p);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

} // namespace

// ---- PCG session API (used by mi_pcg_solve and by bench.py) ----------------
extern "C" int mi_pcg_begin(mi_matrix_t m, const double* psi0, const double* source,
                            const mi_solver_controls* ctl, int precond, int32_t history_len)
{
    if (!m || !psi0 || !source || !ctl) return fail(MI_ERR_ARG, "mi_pcg_begin: bad argument");
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound");
    if (comm_attached(m)) return fail(MI_ERR_STATE, "mi_pcg_begin: a communicator is attached; use mi_pcg_solve or the mi_dpcg_* session");
    if (m->addr->ctx->session && m->addr->ctx->session != m)
        return fail(MI_ERR_STATE, "mi_pcg_begin: another matrix has a PCG session open on this context (one session per context: it owns the solver scratch)");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    hipStream_t s = a->ctx->stream;
    double *psi, *src, *pA, *wA, *rA;
    MICHK(m->vec(3, &psi)); MICHK(m->vec(4, &src)); MICHK(m->vec(5, &pA)); MICHK(m->vec(6, &wA)); MICHK(m->vec(7, &rA));
    k_gather_perm<<<RG, RB, 0, s>>>(psi0, a->perm(), psi, a->L.nCells);
    k_gather_perm<<<RG, RB, 0, s>>>(source, a->perm(), src, a->L.nCells);
    MICHK(solve_prologue(m, ctl, psi, src, wA, rA, pA, history_len));
    // wArA partials of iteration 0 (later iterations get them from k_pcg_update_psi_r)
    if (precond == MI_PRECOND_DIAGONAL) {
        MICHK(ensure_rD(m));
        k_pcg_precond_dot<true><<<RG, RB, 0, s>>>(a->ctx->state.p, m->rD.p, rA, wA, a->L.nCells, a->ctx->partial.p);
    } else if (precond == MI_PRECOND_NONE) {
        k_pcg_precond_dot<false><<<RG, RB, 0, s>>>(a->ctx->state.p, nullptr, rA, wA, a->L.nCells, a->ctx->partial.p);
    }
    HIPCHK(hipGetLastError());
    m->pcgIt = 0; m->pcgPrecond = precond; m->pcgActive = true; m->pcgPReady = false; a->ctx->session = m;
    return MI_OK;
}

extern "C" int mi_pcg_iterate_sampled(mi_matrix_t m, int32_t n_iters, int32_t event_stride, float* amul_ms_sum);
extern "C" int mi_pcg_iterate(mi_matrix_t m, int32_t n_iters, float* amul_ms_sum)
{
    if (!m || !m->pcgActive || n_iters < 0) return fail(MI_ERR_STATE, "mi_pcg_iterate: no active PCG session");
    HIPCHK(hipSetDevice(m->addr->ctx->device));
    return mi_pcg_iterate_sampled(m, n_iters, amul_ms_sum ? 1 : 0, amul_ms_sum);
}

// the same, with HIP events around every event_stride-th Amul only (an event record costs a few us on the stream, which
// matters to a caller timing the whole loop); *amul_ms_sum = mean sampled duration x n_iters
extern "C" int mi_pcg_iterate_sampled(mi_matrix_t m, int32_t n_iters, int32_t event_stride, float* amul_ms_sum)
{
    if (!m || !m->pcgActive || n_iters < 0 || event_stride < 0) return fail(MI_ERR_STATE, "mi_pcg_iterate: no active PCG session");
    HIPCHK(hipSetDevice(m->addr->ctx->device));
    const int stride = amul_ms_sum ? (event_stride > 0 ? event_stride : 1) : 0;
    if (!amul_ms_sum && pcg_persist_usable(m, m->pcgPrecond)) { // small matrix: the whole batch is ONE cooperative launch (persist.inc)
        const int rcP = pcg_persist_enqueue(m, n_iters, m->pcgPrecond);
        if (rcP != MI_ERR_UNSUPPORTED) {
            MICHK(rcP);
            m->pcgIt += n_iters;
            return MI_OK;
        }   // (the runtime refused the cooperative grid; nothing ran: the five launches below take over, here and from now on)
    }
    MICHK(pcg_enqueue(m, m->pcgIt, n_iters, m->pcgPrecond, stride));
    m->pcgIt += n_iters;
    if (amul_ms_sum) {
        HIPCHK(hipStreamSynchronize(m->addr->ctx->stream));
        const int ns = (n_iters + stride - 1) / stride;
        float tot = 0;
        for (int i = 0; i < ns; ++i) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, m->evPool[(size_t)2 * i], m->evPool[(size_t)2 * i + 1])); tot += ms; }
        *amul_ms_sum = ns > 0 ? tot * ((float)n_iters / (float)ns) : 0.f;
    }
    return MI_OK;
}

extern "C" int mi_pcg_end(mi_matrix_t m, double* psi_out, mi_solver_perf* perf, double* hist_host, int32_t hist_len)
{
    if (!m || !m->pcgActive) return fail(MI_ERR_STATE, "mi_pcg_end: no active PCG session");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    double* psi;
    MICHK(m->vec(3, &psi));
    MICHK(pcg_flush(m));
    if (psi_out) k_scatter_perm<<<RG, RB, 0, a->ctx->stream>>>(psi, a->perm(), psi_out, a->L.nCells);
    HIPCHK(hipGetLastError());
    MICHK(fetch_state(a->ctx));
    if (perf) fill_perf(*a->ctx->hostState, perf);
    m->pcgActive = false; a->ctx->session = nullptr; // closed even if the history copy below fails
    MICHK(copy_hist(m, hist_host, hist_len, a->ctx->hostState->nIterations));
    return MI_OK;
}

extern "C" int mi_pcg_solve(mi_matrix_t m, double* psi, const double* source, const mi_solver_controls* ctl,
                            int precond, mi_solver_perf* perf, double* hist_host, int32_t hist_len)
{
    if (!m || !psi || !source || !ctl) return fail(MI_ERR_ARG, "mi_pcg_solve: bad argument");
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound");
    if (comm_attached(m)) return pcg_solve_attached(m, psi, source, ctl, precond, perf, hist_host, hist_len);
    const int histLen = ctl->maxIter + 2;
    if (m->addr->ctx->session) return fail(MI_ERR_STATE, "mi_pcg_solve: a PCG session (mi_pcg_begin) is active on this context; call mi_pcg_end first");
    MICHK(mi_pcg_begin(m, psi, source, ctl, precond, histLen));
    mi_ctx_s* c = m->addr->ctx;
    struct SessionGuard { mi_matrix_s* m; ~SessionGuard() { if (m->pcgActive) { m->pcgActive = false; m->addr->ctx->session = nullptr; } } } sessionGuard{m};
    MICHK(fetch_state(c));
    const int batch = m->addr->ctx->pcgBatch;
    const int limit = ctl->maxIter + (ctl->minIter > ctl->maxIter ? ctl->minIter : 0);
    // Launch-bound regime (small meshes: a 32^3 cavity iteration is five ~5 us launches): one batch of iterations is
    // captured ONCE into a hipGraph -- the iteration counter lives in PcgState, so the kernel arguments never change --
    // and replayed until the device reports done.  Same kernels, same order: results are bit-identical.
    const int wantGraph = c->pcgGraph;
    const bool useGraph = !pcg_persist_usable(m, precond) && (precond == MI_PRECOND_DIAGONAL || precond == MI_PRECOND_NONE) && !c->fuseFinal &&
                          (wantGraph == 1 || (wantGraph < 0 && m->addr->L.nCells <= 4000000));
    if (useGraph && !c->hostState->done) {
        double* psiE; MICHK(m->vec(3, &psiE));
        auto& K = m->pcgGraphKey;
        if (!m->pcgGraph || K.precond != precond || K.batch != batch || K.histLen != m->histLen || K.hist != m->hist.p || K.psi != psiE) {
            if (m->pcgGraph) { (void)hipGraphExecDestroy(m->pcgGraph); m->pcgGraph = nullptr; }
            if (precond != MI_PRECOND_NONE) MICHK(ensure_rD(m));                                  // no allocation inside a capture
            if (m->tilePartial.n < (size_t)m->addr->L.nTiles) MICHK(m->tilePartial.alloc((size_t)m->addr->L.nTiles));
            hipGraph_t g = nullptr;
            HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            const int rcE = pcg_enqueue(m, -1, batch, precond, 0);
            const hipError_t eE = hipStreamEndCapture(c->stream, &g);
            MICHK(rcE);
            HIPCHK(eE);
            const hipError_t eI = hipGraphInstantiate(&m->pcgGraph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            HIPCHK(eI);
            K.precond = precond; K.batch = batch; K.histLen = m->histLen; K.hist = m->hist.p; K.psi = psiE;
        }
        while (!c->hostState->done && m->pcgIt <= limit) {
            HIPCHK(hipGraphLaunch(m->pcgGraph, c->stream));
            m->pcgIt += batch;
            MICHK(fetch_state(c));
        }
        return mi_pcg_end(m, psi, perf, hist_host, hist_len);
    }
    // bodies run for it = 0 .. maxIter inclusive at most (nIterations++ < maxIter, PCG.C:197-204)
    // batches grow 2, 4, 8 ... batch: a solve that converges in a few iterations (every momentum predictor) does not pay
    // for a full batch of launches that exit at their first instruction; a long solve polls `done` 3 times more in total
    int nb = batch < 2 ? batch : 2;
    while (!c->hostState->done && m->pcgIt <= limit) {
        MICHK(mi_pcg_iterate(m, nb, nullptr));
        MICHK(fetch_state(c));
        nb = nb * 2 > batch ? batch : nb * 2;
    }
    return mi_pcg_end(m, psi, perf, hist_host, hist_len);
}


// ---------------------------------------------------------------------------
// distributed PCG: the same device-resident pipeline cut into phases at the
// points where the reference calls Foam::reduce / exchanges processor-patch
// values (PCG.C:142,166,195 -> allReduceTemplates.C:195-208; Amul ->
// lduMatrixUpdateMatrixInterfaces.C:30-276).  The caller (one rank per GPU)
// owns the engine-order vectors and the 8-double scalar block, runs the
// collectives (RCCL) on them between phases, and receives halo values straight
// into pA[n_cells..n_cells+n_ext).  scal: [0] sum wA.rA  [1] sum|rA|  [2] sum wA.pA
// [3] sum psi  [4] normFactor sum.
// ---------------------------------------------------------------------------

extern "C" int mi_dpcg_set_buffers(mi_matrix_t m, double* psi_e, double* src_e, double* pA_e, double* wA_e, double* rA_e,
                                   double* scal8, double* send_buf, const mi_solver_controls* ctl, int precond, int32_t history_len)
{
    if (!m || !psi_e || !src_e || !pA_e || !wA_e || !rA_e || !scal8 || !ctl) return fail(MI_ERR_ARG, "mi_dpcg_set_buffers: bad argument");
    if (precond != MI_PRECOND_DIAGONAL && precond != MI_PRECOND_NONE) return fail(MI_ERR_ARG, "distributed PCG supports the diagonal / none preconditioners");
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound");
    if (comm_any_ami(m) || comm_any_factor(m))
        return fail(MI_ERR_UNSUPPORTED, "the phase-split distributed PCG does not interpolate cyclicAMI / transformed patches: use mi_pcg_solve on the attached matrix");
    mi_ctx_s* c = m->addr->ctx;
    HIPCHK(hipSetDevice(c->device));
    m->dp.psi = psi_e; m->dp.src = src_e; m->dp.pA = pA_e; m->dp.wA = wA_e; m->dp.rA = rA_e; m->dp.scal = scal8; m->dp.send = send_buf; m->dp.precond = precond;
    const int histLen = history_len > 0 ? history_len : 1;
    if (m->hist.n < (size_t)histLen) MICHK(m->hist.alloc((size_t)histLen));
    m->histLen = history_len;
    if (m->tilePartial.n < (size_t)m->addr->L.nTiles) MICHK(m->tilePartial.alloc((size_t)m->addr->L.nTiles));
    PcgState h; memset(&h, 0, sizeof(h));
    h.tolerance = ctl->tolerance; h.relTol = ctl->relTol; h.maxIter = ctl->maxIter; h.minIter = ctl->minIter;
    *c->hostState = h;
    HIPCHK(hipMemcpyAsync(c->state.p, c->hostState, sizeof(PcgState), hipMemcpyHostToDevice, c->stream));
    if (precond == MI_PRECOND_DIAGONAL) MICHK(ensure_rD(m));
    return MI_OK;
}

// phases: see include/mi_ldu.h
extern "C" int mi_dpcg_phase(mi_matrix_t m, int phase, int32_t it, double arg)
{
    if (!m || !m->dp.psi) return fail(MI_ERR_STATE, "mi_dpcg_phase: call mi_dpcg_set_buffers first");
    mi_addr_s* a = m->addr;
    mi_ctx_s* c = a->ctx;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int64_t n = a->L.nCells;
    const mi_dpcg_s& B = m->dp;
    double* P = c->partial.p; // scratch partial slots
    auto finalize = [&](const double* partial, double* dst) { k_reduce_final<<<1, RB, 0, s>>>(partial, dst); };
    switch (phase) {
    case 0: // prologue: gather psi at the coupled patches for the exchange
        MICHK(mi_halo_pack_engine(a, B.psi, B.send));
        break;
    case 1: // wA = A psi ; rA = src - wA ; sumA ; local sum(psi)
        MICHK(launch_tile<OP_AMUL>(m, false, B.psi, nullptr, nullptr, B.wA, 0.0, 0));
        k_sub<<<RG, RB, 0, s>>>(B.rA, B.src, B.wA, n);
        MICHK(launch_tile<OP_SUMA>(m, false, nullptr, nullptr, nullptr, B.pA, 0.0, 0));
        k_reduce<RED_SUM><<<RG, RB, 0, s>>>(B.psi, nullptr, n, P);
        finalize(P, B.scal + 3);
        break;
    case 2: // arg = global average of psi: normFactor partial, sum|rA|, first wArA
        k_normfactor<<<RG, RB, 0, s>>>(B.wA, B.src, B.pA, arg, n, P);
        finalize(P, B.scal + 4);
        k_reduce<RED_MAG><<<RG, RB, 0, s>>>(B.rA, nullptr, n, P + RG);
        finalize(P + RG, B.scal + 1);
        if (B.precond == MI_PRECOND_DIAGONAL) k_pcg_precond_dot<true><<<RG, RB, 0, s>>>(c->state.p, m->rD.p, B.rA, B.wA, n, P + 2 * RG);
        else k_pcg_precond_dot<false><<<RG, RB, 0, s>>>(c->state.p, nullptr, B.rA, B.wA, n, P + 2 * RG);
        finalize(P + 2 * RG, B.scal + 0);
        break;
    case 3: // after the allreduce of scal[0,1,4]
        k_solve_init<true><<<1, RB, 0, s>>>(c->state.p, B.scal + 4, B.scal + 1, m->hist.p, m->histLen);
        break;
    case 10: // iteration `it`: (residual test of it-1,) pA update, pack pA for the exchange
        // the residual test of iteration it-1 (PCG.C:195-204) rides at the head of the pA update, like on one GPU
        if (B.precond == MI_PRECOND_DIAGONAL) k_pcg_update_p<1, true><<<RG, RB, 0, s>>>(c->state.p, it, B.scal + 0, nullptr, m->rD.p, B.rA, B.pA, n, B.scal + 1, m->hist.p, m->histLen);
        else k_pcg_update_p<2, true><<<RG, RB, 0, s>>>(c->state.p, it, B.scal + 0, nullptr, nullptr, B.rA, B.pA, n, B.scal + 1, m->hist.p, m->histLen);
        MICHK(mi_halo_pack_engine(a, B.pA, B.send));
        break;
    case 11: // interior tiles: overlap with the halo exchange
        MICHK(launch_tile<OP_AMUL>(m, false, B.pA, nullptr, nullptr, B.wA, 0.0, 1, m->tilePartial.p));
        break;
    case 12: // boundary tiles (halo has arrived) + local sum wA.pA
        MICHK(launch_tile<OP_AMUL>(m, false, B.pA, nullptr, nullptr, B.wA, 0.0, 2, m->tilePartial.p + a->nInterior));
        k_fold_final<<<1, 1024, 0, s>>>(m->tilePartial.p, a->L.nTiles, B.scal + 2);
        break;
    case 13: // after the allreduce of scal[2]: psi, rA updates; local sum|rA| and next wArA
        if (B.precond == MI_PRECOND_DIAGONAL)
            k_pcg_update_psi_r<1, true><<<RG, RB, 0, s>>>(c->state.p, it, B.scal + 2, B.pA, B.wA, m->rD.p, B.psi, B.rA, n, P + RG, P + 2 * RG);
        else
            k_pcg_update_psi_r<2, true><<<RG, RB, 0, s>>>(c->state.p, it, B.scal + 2, B.pA, B.wA, nullptr, B.psi, B.rA, n, P + RG, P + 2 * RG);
        k_reduce_final2<<<2, RB, 0, s>>>(P + RG, B.scal + 1, P + 2 * RG, B.scal + 0);
        break;
    case 14: // residual test of the last enqueued iteration (after the allreduce of scal[0,1])
        k_pcg_final<true><<<1, RB, 0, s>>>(c->state.p, it, B.scal + 1, m->hist.p, m->histLen);
        break;
    default:
        return fail(MI_ERR_ARG, "mi_dpcg_phase: unknown phase");
    }
    HIPCHK(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_dpcg_status(mi_matrix_t m, mi_solver_perf* perf, int32_t* done, double* hist_host, int32_t hist_len)
{
    if (!m || !perf) return fail(MI_ERR_ARG, "mi_dpcg_status: bad argument");
    mi_ctx_s* c = m->addr->ctx;
    HIPCHK(hipSetDevice(c->device));
    MICHK(dpcg_flush_if_fused(m));   // the fused loop defers psi += alpha pA to the next p-update: add what the last iteration owes
    MICHK(fetch_state(c));
    MICHK(peer_check(m));
    fill_perf(*c->hostState, perf);
    if (done) *done = c->hostState->done;
    MICHK(copy_hist(m, hist_host, hist_len, c->hostState->nIterations));
    return MI_OK;
}

// ---- host-synchronous Krylov solvers for the asymmetric path -----------------
namespace {
struct HostPerf {
    double initialResidual = 0, finalResidual = 0, normFactor = 0;
    int nIterations = 0, converged = 0, singular = 0;
    double tolerance, relTol; int maxIter, minIter;
    bool checkConvergence()
    {
        converged = (finalResidual < tolerance) || (relTol > SP_SMALL && finalResidual < relTol * initialResidual);
        return converged != 0;
    }
    bool checkSingularity(double v) { singular = (v < SP_VSMALL); return singular != 0; }
};

int finish_host(mi_matrix_s* m, const HostPerf& hp, const Table<double>& hist, double* psi_e, double* psi_out,
                mi_solver_perf* perf, double* hist_host, int hist_len)
{
    mi_addr_s* a = m->addr;
    k_scatter_perm<<<RG, RB, 0, a->ctx->stream>>>(psi_e, a->perm(), psi_out, a->L.nCells);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(a->ctx->stream));
    if (perf) {
        perf->initialResidual = hp.initialResidual; perf->finalResidual = hp.finalResidual; perf->normFactor = hp.normFactor;
        perf->nIterations = hp.nIterations; perf->converged = hp.converged; perf->singular = hp.singular; perf->reserved = 0;
    }
    if (hist_host) for (int i = 0; i < hist_len && i < (int)hist.size(); ++i) hist_host[i] = hist[(size_t)i];
    return MI_OK;
}

int host_prologue(mi_matrix_s* m, const mi_solver_controls* ctl, const double* psi_e, const double* src_e,
                  double* wA, double* rA, double* tmp, HostPerf& hp, Table<double>& hist)
{
    MICHK(solve_prologue(m, ctl, psi_e, src_e, wA, rA, tmp, 1));
    MICHK(fetch_state(m->addr->ctx));
    const PcgState& h = *m->addr->ctx->hostState;
    hp.tolerance = ctl->tolerance; hp.relTol = ctl->relTol; hp.maxIter = ctl->maxIter; hp.minIter = ctl->minIter;
    hp.normFactor = h.normFactor; hp.initialResidual = h.initialResidual; hp.finalResidual = h.finalResidual;
    hist.clear(); hist.push_back(hp.initialResidual);
    return MI_OK;
}
} // namespace

namespace {
// y0 = Op x0, y1 = Op^T x1 in ONE pass over the coefficients (multi.inc: tile_kernel_multi with one component), Op = A or the
// AINV apply; gate: &PcgState::done of the running solve or nullptr
int tile_pair(mi_matrix_s* m, bool ainv, const double* x0, const double* x1, double* y0, double* y1, const int32_t* gate, double* dotPartial, bool* fused);
int multi_exchange(mi_matrix_s* m, int nv, double* const* xs);   // multi.inc: one halo exchange for several operand vectors
// enqueue PBiCG iteration bodies it0 .. it0+count-1 (no host sync): precondition both residuals (+ fused sum wA.rT),
// update pA/pT, Amul, Tmul, sum wA.pT, update psi/rA/rT (+ sum|rA|), convergence test
int bicg_enqueue(mi_matrix_s* m, int it0, int count, int precond, double* psi, double* pA, double* wA, double* rA,
                 double* pT, double* wT, double* rT)
{
    mi_addr_s* a = m->addr;
    mi_ctx_s* c = a->ctx;
    hipStream_t s = c->stream;
    const int64_t n = a->L.nCells;
    double* P1 = c->partial.p; double* P2 = c->partial.p + RG; double* P3 = c->partial.p + 2 * RG;
    if (precond != MI_PRECOND_NONE) MICHK(ensure_rD(m));
    if (m->tilePartial.n < (size_t)a->L.nTiles) MICHK(m->tilePartial.alloc((size_t)a->L.nTiles));
    struct Gate { mi_matrix_s* m; explicit Gate(mi_matrix_s* mm) : m(mm) { m->gateDone = true; } ~Gate() { m->gateDone = false; } } gate(m);
    for (int it = it0; it < it0 + count; ++it) {
        // the plain and the transposed pass share one staging of the coefficients; on a decomposed case (round 4) pA and pT also
        // share ONE halo exchange (cyclicAMI patches are interpolated per operand inside tile_op: they keep the separate passes)
        const bool paired = c->pairAT && (!comm_attached(m) || (!comm_any_ami(m) && !comm_any_compact(m)));
        if (precond == MI_PRECOND_AINV) {
            bool fusedDot = false;   // sum wA.rT out of the preconditioner pass (per-tile partials, folded like the Amul's in PCG)
            if (paired) MICHK(tile_pair(m, true, rA, rT, wA, wT, &c->state.p->done, m->tilePartial.p, &fusedDot));
            else {
            MICHK(launch_tile<OP_AINV>(m, false, rA, nullptr, m->rD.p, wA, 0.0, 0));
            MICHK(launch_tile<OP_AINV>(m, true, rT, nullptr, m->rD.p, wT, 0.0, 0));
            }
            if (fusedDot) k_fold_partials<<<1, 1024, 0, s>>>(m->tilePartial.p, a->L.nTiles, P1);
            else k_reduce<RED_PROD><<<RG, RB, 0, s>>>(wA, rT, n, P1);
        } else if (precond == MI_PRECOND_DIAGONAL) k_bicg_precond_dot<true><<<RG, RB, 0, s>>>(c->state.p, m->rD.p, rA, rT, wA, wT, n, P1);
        else k_bicg_precond_dot<false><<<RG, RB, 0, s>>>(c->state.p, nullptr, rA, rT, wA, wT, n, P1);
        MICHK(globalize(m, P1));
        k_bicg_update_p<<<RG, RB, 0, s>>>(c->state.p, it, P1, wA, wT, pA, pT, n);
        bool fusedDot2 = false;      // sum wA.pT out of the Amul / Tmul pass
        if (paired) {
            if (comm_attached(m)) { double* ops[2] = {pA, pT}; MICHK(multi_exchange(m, 2, ops)); }
            MICHK(tile_pair(m, false, pA, pT, wA, wT, &c->state.p->done, m->tilePartial.p, &fusedDot2));
        } else {
        MICHK(tile_op<OP_AMUL>(m, false, pA, nullptr, nullptr, wA, 0.0));   // halo exchange inside when attached
        MICHK(tile_op<OP_AMUL>(m, true, pT, nullptr, nullptr, wT, 0.0));
        }
        if (fusedDot2) k_fold_partials<<<1, 1024, 0, s>>>(m->tilePartial.p, a->L.nTiles, P2);
        else k_reduce<RED_PROD><<<RG, RB, 0, s>>>(wA, pT, n, P2);
        MICHK(globalize(m, P2));
        k_bicg_update_psi_r<<<RG, RB, 0, s>>>(c->state.p, it, P2, pA, wA, wT, psi, rA, rT, n, P3);
        MICHK(globalize(m, P3));
        k_pcg_final<false><<<1, RB, 0, s>>>(c->state.p, it, P3, m->hist.p, m->histLen);
    }
    HIPCHK(hipGetLastError());
    return MI_OK;
}

// PBiCG with every scalar on the device (single GPU; a communicator-attached matrix keeps the host-stepped loop below,
// whose sums are all-reduced)
int pbicg_solve_device(mi_matrix_s* m, double* psi_io, const double* source, const mi_solver_controls* ctl, int precond,
                       mi_solver_perf* perf, double* hist_host, int32_t hist_len)
{
    mi_addr_s* a = m->addr;
    mi_ctx_s* c = a->ctx;
    hipStream_t s = c->stream;
    const int64_t n = a->L.nCells;
    double *psi, *src, *pA, *wA, *rA, *pT, *wT, *rT;
    MICHK(m->vec(3, &psi)); MICHK(m->vec(4, &src)); MICHK(m->vec(5, &pA)); MICHK(m->vec(6, &wA)); MICHK(m->vec(7, &rA));
    MICHK(m->vec(8, &pT)); MICHK(m->vec(9, &wT)); MICHK(m->vec(10, &rT));
    k_gather_perm<<<RG, RB, 0, s>>>(psi_io, a->perm(), psi, a->L.nCells);
    k_gather_perm<<<RG, RB, 0, s>>>(source, a->perm(), src, a->L.nCells);
    const int histLen = ctl->maxIter + 2;
    MICHK(solve_prologue(m, ctl, psi, src, wA, rA, pA, histLen));     // wA = A psi, rA = src - wA, normFactor, first test
    MICHK(tile_op<OP_AMUL>(m, true, psi, nullptr, nullptr, wT, 0.0));
    k_sub<<<RG, RB, 0, s>>>(rT, src, wT, n);
    MICHK(fetch_state(c));
    const int batch = m->addr->ctx->pcgBatch;
    int it = 0, nb = batch < 2 ? batch : 2;   // growing batches, as in mi_pcg_solve
    while (!c->hostState->done && it <= ctl->maxIter + (ctl->minIter > ctl->maxIter ? ctl->minIter : 0)) {
        MICHK(bicg_enqueue(m, it, nb, precond, psi, pA, wA, rA, pT, wT, rT));
        it += nb;
        MICHK(fetch_state(c));
        MICHK(peer_check(m));
        nb = nb * 2 > batch ? batch : nb * 2;
    }
    k_scatter_perm<<<RG, RB, 0, s>>>(psi, a->perm(), psi_io, a->L.nCells);
    HIPCHK(hipGetLastError());
    if (perf) fill_perf(*c->hostState, perf);
    MICHK(copy_hist(m, hist_host, hist_len, c->hostState->nIterations));
    HIPCHK(hipStreamSynchronize(s));
    return MI_OK;
}
} // namespace

extern "C" int mi_pbicg_solve_multi(mi_matrix_t m, int32_t nrhs, const double* const* diag_dev, double* const* psi_io, const double* const* source,
                                    const mi_solver_controls* ctl, int precond, mi_solver_perf* perf, double* hist_host, int32_t hist_len);
namespace {
bool pbicg_single_via_multi(const mi_matrix_s* m);   // multi.inc
}
extern "C" int mi_pbicg_solve(mi_matrix_t m, double* psi_io, const double* source, const mi_solver_controls* ctl,
                              int precond, mi_solver_perf* perf, double* hist_host, int32_t hist_len)
{
    if (!m || !psi_io || !source || !ctl) return fail(MI_ERR_ARG, "mi_pbicg_solve: bad argument");
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound");
    if (m->addr->ctx->session) return fail(MI_ERR_STATE, "mi_pbicg_solve: a PCG session (mi_pcg_begin) is active on this context; call mi_pcg_end first");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    if (m->addr->ctx->pbicgHostStepped == 0) { // also with a communicator attached: the loop all-reduces its sums on the device (globalize)
        // round 6: one right-hand side through the multi-vector solver -- its prologue forms A psi, A^T psi and both residuals in ONE pass over
        // the coefficients (here: two tile passes and two vector passes; a one-iteration energy-equation solve spent half its time there),
        // its iterations are the same pipelined passes.  Same arithmetic per operand.  (mi_pbicg_solve_multi hands matrices it cannot take
        // -- attached with cyclicAMI patches / compact entries, MI_PBICG_MULTI=0 -- back to pbicg_solve_device below.)
        if (pbicg_single_via_multi(m)) return mi_pbicg_solve_multi(m, 1, nullptr, &psi_io, &source, ctl, precond, perf, hist_host, hist_len);
        return pbicg_solve_device(m, psi_io, source, ctl, precond, perf, hist_host, hist_len);
    }
    hipStream_t s = a->ctx->stream;
    const int64_t n = a->L.nCells;
    double *psi, *src, *pA, *wA, *rA, *pT, *wT, *rT;
    MICHK(m->vec(3, &psi)); MICHK(m->vec(4, &src)); MICHK(m->vec(5, &pA)); MICHK(m->vec(6, &wA)); MICHK(m->vec(7, &rA));
    MICHK(m->vec(8, &pT)); MICHK(m->vec(9, &wT)); MICHK(m->vec(10, &rT));
    k_gather_perm<<<RG, RB, 0, s>>>(psi_io, a->perm(), psi, a->L.nCells);
    k_gather_perm<<<RG, RB, 0, s>>>(source, a->perm(), src, a->L.nCells);
    HostPerf hp; Table<double> hist;
    MICHK(host_prologue(m, ctl, psi, src, wA, rA, pA, hp, hist));
    MICHK(tile_op<OP_AMUL>(m, true, psi, nullptr, nullptr, wT, 0.0));
    k_sub<<<RG, RB, 0, s>>>(rT, src, wT, n);
    double wArT = SP_GREAT, wArTold = wArT;
    if (hp.minIter > 0 || !hp.checkConvergence()) {
        do {
            wArTold = wArT;
            MICHK(precond_engine(m, precond, false, rA, wA));
            MICHK(precond_engine(m, precond, true, rT, wT));
            MICHK(reduce_sync<RED_PROD>(m, wA, rT, &wArT));
            if (hp.nIterations == 0) {
                HIPCHK(hipMemcpyAsync(pA, wA, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpyAsync(pT, wT, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
            } else {
                const double beta = wArT / wArTold;
                k_xpsy<<<RG, RB, 0, s>>>(pA, wA, beta, pA, n);
                k_xpsy<<<RG, RB, 0, s>>>(pT, wT, beta, pT, n);
            }
            MICHK(tile_op<OP_AMUL>(m, false, pA, nullptr, nullptr, wA, 0.0));
            MICHK(tile_op<OP_AMUL>(m, true, pT, nullptr, nullptr, wT, 0.0));
            double wApT = 0;
            MICHK(reduce_sync<RED_PROD>(m, wA, pT, &wApT));
            if (hp.checkSingularity(fabs(wApT) / hp.normFactor)) break;
            const double alpha = wArT / wApT;
            k_xpsy<<<RG, RB, 0, s>>>(psi, psi, alpha, pA, n);
            k_xpsy<<<RG, RB, 0, s>>>(rA, rA, -alpha, wA, n);
            k_xpsy<<<RG, RB, 0, s>>>(rT, rT, -alpha, wT, n);
            double sm = 0;
            MICHK(reduce_sync<RED_MAG>(m, rA, nullptr, &sm));
            hp.finalResidual = sm / hp.normFactor;
            hist.push_back(hp.finalResidual);
        } while ((hp.nIterations++ < hp.maxIter && !hp.checkConvergence()) || hp.nIterations < hp.minIter);
    }
    return finish_host(m, hp, hist, psi, psi_io, perf, hist_host, hist_len);
}

namespace {
int stab_enqueue(mi_matrix_s* m, int it0, int count, int precond, bool quirk, double* psi, double* pA, double* yA, double* rA,
                 double* AyA, double* sA, double* zA, double* tA, double* rA0)
{
    mi_addr_s* a = m->addr;
    mi_ctx_s* c = a->ctx;
    hipStream_t s = c->stream;
    const int64_t n = a->L.nCells;
    double* P1 = c->partial.p; double* P2 = c->partial.p + RG; double* P3 = c->partial.p + 2 * RG; double* P4 = c->partial.p + 3 * RG;
    struct Gate { mi_matrix_s* m; explicit Gate(mi_matrix_s* mm) : m(mm) { m->gateDone = true; } ~Gate() { m->gateDone = false; } } gate(m);
    for (int it = it0; it < it0 + count; ++it) {
        k_reduce<RED_PROD><<<RG, RB, 0, s>>>(rA0, rA, n, P1);
        MICHK(globalize(m, P1));
        k_stab_update_p<<<RG, RB, 0, s>>>(c->state.p, it, P1, rA, AyA, pA, n);
        MICHK(precond_engine(m, precond, false, pA, yA));
        MICHK(tile_op<OP_AMUL>(m, false, yA, nullptr, nullptr, AyA, 0.0));   // halo exchange inside when attached
        k_reduce<RED_PROD><<<RG, RB, 0, s>>>(rA0, AyA, n, P2);
        MICHK(globalize(m, P2));
        k_stab_s<<<RG, RB, 0, s>>>(c->state.p, it, P2, rA, AyA, sA, n, P3);
        MICHK(globalize(m, P3));
        k_stab_mid<<<RG, RB, 0, s>>>(c->state.p, P3, yA, psi, n);
        k_stab_mid_final<<<1, RB, 0, s>>>(c->state.p, it, P3, m->hist.p, m->histLen);
        MICHK(precond_engine(m, precond, false, sA, zA));
        MICHK(tile_op<OP_AMUL>(m, false, zA, nullptr, nullptr, tA, 0.0));
        k_reduce_two<<<RG, RB, 0, s>>>(tA, sA, n, P1, P2);
        MICHK(globalize(m, P1, P2));
        k_stab_update<<<RG, RB, 0, s>>>(c->state.p, P1, P2, yA, quirk ? yA : zA, sA, tA, psi, rA, n, P4);
        MICHK(globalize(m, P4));
        k_pcg_final<false><<<1, RB, 0, s>>>(c->state.p, it, P4, m->hist.p, m->histLen);
    }
    HIPCHK(hipGetLastError());
    return MI_OK;
}

// PBiCGStab with every scalar on the device (single GPU; communicator-attached matrices keep the host-stepped loop)
int pbicgstab_solve_device(mi_matrix_s* m, double* psi_io, const double* source, const mi_solver_controls* ctl, int precond,
                           int replicate_quirk, mi_solver_perf* perf, double* hist_host, int32_t hist_len)
{
    mi_addr_s* a = m->addr;
    mi_ctx_s* c = a->ctx;
    hipStream_t s = c->stream;
    const int64_t n = a->L.nCells;
    if (c->partial.n < (size_t)4 * RG) return fail(MI_ERR_STATE, "partial buffer too small");
    double *psi, *src, *pA, *yA, *rA, *AyA, *sA, *zA, *tA, *rA0;
    MICHK(m->vec(3, &psi)); MICHK(m->vec(4, &src)); MICHK(m->vec(5, &pA)); MICHK(m->vec(6, &yA)); MICHK(m->vec(7, &rA));
    MICHK(m->vec(8, &AyA)); MICHK(m->vec(9, &sA)); MICHK(m->vec(10, &zA)); MICHK(m->vec(11, &tA)); MICHK(m->vec(12, &rA0));
    k_gather_perm<<<RG, RB, 0, s>>>(psi_io, a->perm(), psi, a->L.nCells);
    k_gather_perm<<<RG, RB, 0, s>>>(source, a->perm(), src, a->L.nCells);
    if (precond != MI_PRECOND_NONE) MICHK(ensure_rD(m));
    const int histLen = ctl->maxIter + 2;
    MICHK(solve_prologue(m, ctl, psi, src, yA, rA, pA, histLen));
    HIPCHK(hipMemcpyAsync(rA0, rA, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    MICHK(fetch_state(c));
    const int batch = m->addr->ctx->pcgBatch;
    int it = 0, nb = batch < 2 ? batch : 2;   // growing batches, as in mi_pcg_solve
    while (!c->hostState->done && it <= ctl->maxIter + (ctl->minIter > ctl->maxIter ? ctl->minIter : 0)) {
        MICHK(stab_enqueue(m, it, nb, precond, replicate_quirk != 0, psi, pA, yA, rA, AyA, sA, zA, tA, rA0));
        it += nb;
        MICHK(fetch_state(c));
        MICHK(peer_check(m));
        nb = nb * 2 > batch ? batch : nb * 2;
    }
    k_scatter_perm<<<RG, RB, 0, s>>>(psi, a->perm(), psi_io, a->L.nCells);
    HIPCHK(hipGetLastError());
    if (perf) fill_perf(*c->hostState, perf);
    MICHK(copy_hist(m, hist_host, hist_len, c->hostState->nIterations));
    HIPCHK(hipStreamSynchronize(s));
    return MI_OK;
}
} // namespace

extern "C" int mi_pbicgstab_solve(mi_matrix_t m, double* psi_io, const double* source, const mi_solver_controls* ctl,
                                  int precond, int replicate_quirk, mi_solver_perf* perf, double* hist_host, int32_t hist_len)
{
    if (!m || !psi_io || !source || !ctl) return fail(MI_ERR_ARG, "mi_pbicgstab_solve: bad argument");
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound");
    if (m->addr->ctx->session) return fail(MI_ERR_STATE, "mi_pbicgstab_solve: a PCG session (mi_pcg_begin) is active on this context; call mi_pcg_end first");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    if (m->addr->ctx->pbicgHostStepped == 0)
        return pbicgstab_solve_device(m, psi_io, source, ctl, precond, replicate_quirk, perf, hist_host, hist_len);
    hipStream_t s = a->ctx->stream;
    const int64_t n = a->L.nCells;
    double *psi, *src, *pA, *yA, *rA, *AyA, *sA, *zA, *tA, *rA0, *res1;
    MICHK(m->vec(3, &psi)); MICHK(m->vec(4, &src)); MICHK(m->vec(5, &pA)); MICHK(m->vec(6, &yA)); MICHK(m->vec(7, &rA));
    MICHK(m->vec(8, &AyA)); MICHK(m->vec(9, &sA)); MICHK(m->vec(10, &zA)); MICHK(m->vec(11, &tA)); MICHK(m->vec(12, &rA0));
    MICHK(m->vec(13, &res1));
    k_gather_perm<<<RG, RB, 0, s>>>(psi_io, a->perm(), psi, a->L.nCells);
    k_gather_perm<<<RG, RB, 0, s>>>(source, a->perm(), src, a->L.nCells);
    HostPerf hp; Table<double> hist;
    MICHK(host_prologue(m, ctl, psi, src, yA, rA, pA, hp, hist));
    if (hp.minIter > 0 || !hp.checkConvergence()) {
        HIPCHK(hipMemcpyAsync(rA0, rA, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
        double rA0rA = 0, alpha = 0, omega = 0;
        do {
            const double rA0rAold = rA0rA;
            MICHK(reduce_sync<RED_PROD>(m, rA0, rA, &rA0rA));
            if (hp.checkSingularity(fabs(rA0rA))) break;
            if (hp.nIterations == 0) {
                HIPCHK(hipMemcpyAsync(pA, rA, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
            } else {
                if (hp.checkSingularity(fabs(omega))) break;
                const double beta = (rA0rA / rA0rAold) * (alpha / omega);
                k_xpsy<<<RG, RB, 0, s>>>(res1, pA, -omega, AyA, n); // result1 = pA - omega*AyA
                k_xpsy<<<RG, RB, 0, s>>>(pA, rA, beta, res1, n);    // pA = rA + beta*result1
            }
            MICHK(precond_engine(m, precond, false, pA, yA));
            MICHK(tile_op<OP_AMUL>(m, false, yA, nullptr, nullptr, AyA, 0.0));
            double rA0AyA = 0;
            MICHK(reduce_sync<RED_PROD>(m, rA0, AyA, &rA0AyA));
            alpha = rA0rA / rA0AyA;
            k_xpsy<<<RG, RB, 0, s>>>(sA, rA, -alpha, AyA, n);
            double sm = 0;
            MICHK(reduce_sync<RED_MAG>(m, sA, nullptr, &sm));
            hp.finalResidual = sm / hp.normFactor;
            if (hp.checkConvergence()) {
                k_xpsy<<<RG, RB, 0, s>>>(psi, psi, alpha, yA, n);
                hp.nIterations++;
                hist.push_back(hp.finalResidual);
                return finish_host(m, hp, hist, psi, psi_io, perf, hist_host, hist_len);
            }
            MICHK(precond_engine(m, precond, false, sA, zA));
            MICHK(tile_op<OP_AMUL>(m, false, zA, nullptr, nullptr, tA, 0.0));
            double tAtA = 0, tAsA = 0;
            MICHK(reduce_sync<RED_PROD>(m, tA, tA, &tAtA));
            MICHK(reduce_sync<RED_PROD>(m, tA, sA, &tAsA));
            omega = tAsA / tAtA;
            k_xpsy<<<RG, RB, 0, s>>>(psi, psi, alpha, yA, n);
            k_xpsy<<<RG, RB, 0, s>>>(psi, psi, omega, replicate_quirk ? yA : zA, n);
            k_xpsy<<<RG, RB, 0, s>>>(rA, sA, -omega, tA, n);
            MICHK(reduce_sync<RED_MAG>(m, rA, nullptr, &sm));
            hp.finalResidual = sm / hp.normFactor;
            hist.push_back(hp.finalResidual);
        } while ((hp.nIterations++ < hp.maxIter && !hp.checkConvergence()) || hp.nIterations < hp.minIter);
    }
    return finish_host(m, hp, hist, psi, psi_io, perf, hist_host, hist_len);
}

extern "C" int mi_smooth_solve(mi_matrix_t m, double* psi_io, const double* source, const mi_solver_controls* ctl,
                               double omega, int32_t n_sweeps, mi_solver_perf* perf, double* hist_host, int32_t hist_len)
{
    if (!m || !psi_io || !source || !ctl || n_sweeps == 0) return fail(MI_ERR_ARG, "mi_smooth_solve: bad argument");
    if (!m->bound) return fail(MI_ERR_STATE, "matrix coefficients not bound");
    if (m->addr->ctx->session) return fail(MI_ERR_STATE, "mi_smooth_solve: a PCG session (mi_pcg_begin) is active on this context; call mi_pcg_end first");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    hipStream_t s = a->ctx->stream;
    double *psi, *src, *tmp, *wA, *rA, *psi2;
    MICHK(m->vec(3, &psi)); MICHK(m->vec(4, &src)); MICHK(m->vec(5, &tmp)); MICHK(m->vec(6, &wA)); MICHK(m->vec(7, &rA));
    MICHK(m->vec(8, &psi2));
    k_gather_perm<<<RG, RB, 0, s>>>(psi_io, a->perm(), psi, a->L.nCells);
    k_gather_perm<<<RG, RB, 0, s>>>(source, a->perm(), src, a->L.nCells);
    HostPerf hp; Table<double> hist;
    hp.tolerance = ctl->tolerance; hp.relTol = ctl->relTol; hp.maxIter = ctl->maxIter; hp.minIter = ctl->minIter;
    double *cur = psi, *nxt = psi2;
    auto sweeps = [&](int cnt) -> int {
        for (int sw = 0; sw < cnt; ++sw) {
            MICHK(tile_op<OP_JACOBI>(m, false, cur, src, nullptr, nxt, omega));
            double* t = cur; cur = nxt; nxt = t;
        }
        return MI_OK;
    };
    if (n_sweeps < 0) { // smoothSolver.C:87-110
        MICHK(sweeps(-n_sweeps));
        hp.nIterations -= n_sweeps;
    } else {
        MICHK(host_prologue(m, ctl, psi, src, wA, rA, tmp, hp, hist));
        if (hp.minIter > 0 || !hp.checkConvergence()) {
            do {
                MICHK(sweeps(n_sweeps));
                MICHK(tile_op<OP_RESIDUAL>(m, false, cur, src, nullptr, rA, 0.0));
                double sm = 0;
                MICHK(reduce_sync<RED_MAG>(m, rA, nullptr, &sm));
                hp.finalResidual = sm / hp.normFactor;
                hist.push_back(hp.finalResidual);
            } while (((hp.nIterations += n_sweeps) < hp.maxIter && !hp.checkConvergence()) || hp.nIterations < hp.minIter);
        }
    }
    return finish_host(m, hp, hist, cur, psi_io, perf, hist_host, hist_len);
}

// ---------------------------------------------------------------------------
// benchmark hooks
// ---------------------------------------------------------------------------
// diagnostic: resident workgroups per CU of the Amul kernel as the runtime sees it (LDS / VGPR limits)
extern "C" int mi_debug_occupancy(mi_matrix_t m, int32_t* blocks_per_cu, int32_t* lds_bytes_out, int32_t* block_size)
{
    if (!m || !blocks_per_cu) return fail(MI_ERR_ARG, "mi_debug_occupancy: bad argument");
    mi_addr_s* a = m->addr;
    HIPCHK(hipSetDevice(a->ctx->device));
    int32_t o1, o2, o3, o4;
    const size_t lds = lds_bytes(a->L, m->asym, false, &o1, &o2, &o3, &o4);
    const int bs = a->ctx->amulBS ? a->ctx->amulBS : ((lds > 53 * 1024) ? 1024 : (a->L.maxCells <= 256 ? 256 : 512));
    int nb = 0;
#define MI_OCC(BS) { if (a->compact) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)tile_kernel<OP_AMUL, false, false, BS, true>, BS, lds)); \
                     else HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)tile_kernel<OP_AMUL, false, false, BS, false>, BS, lds)); }
    if (bs == 1024) MI_OCC(1024) else if (bs == 512) MI_OCC(512) else MI_OCC(256)
#undef MI_OCC
    *blocks_per_cu = nb;
    if (lds_bytes_out) *lds_bytes_out = (int32_t)lds;
    if (block_size) *block_size = bs;
    return MI_OK;
}

extern "C" int mi_bench_amul(mi_matrix_t m, int32_t reps, float* ms_out)
{
    if (!m || reps <= 0 || !ms_out) return fail(MI_ERR_ARG, "mi_bench_amul: bad argument");
    mi_ctx_s* c = m->addr->ctx;
    HIPCHK(hipSetDevice(c->device));
    double *x, *y;
    MICHK(m->vec(5, &x)); MICHK(m->vec(6, &y));
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; ++i) MICHK(launch_tile<OP_AMUL>(m, false, x, nullptr, nullptr, y, 0.0, 0));
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms_out, c->ev0, c->ev1));
    return MI_OK;
}

extern "C" int mi_bench_pcg_iters(mi_matrix_t m, const double* source, int32_t iters, int precond, float* ms_out, float* amul_ms_out)
{
    if (!m || !source || iters <= 0 || !ms_out) return fail(MI_ERR_ARG, "mi_bench_pcg_iters: bad argument");
    mi_ctx_s* c = m->addr->ctx;
    HIPCHK(hipSetDevice(c->device));
    mi_solver_controls ctl; ctl.tolerance = 0; ctl.relTol = 0; ctl.maxIter = iters + 16; ctl.minIter = 0;
    double* zero;
    MICHK(m->vec(14, &zero));
    // psi0 = 0 in caller order == 0 in engine order
    MICHK(mi_pcg_begin(m, zero, source, &ctl, precond, 1));
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    MICHK(pcg_enqueue(m, 0, iters, precond, 0));
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms_out, c->ev0, c->ev1));
    m->pcgActive = false; c->session = nullptr;
    if (amul_ms_out) { float t = 0; MICHK(mi_bench_amul(m, iters, &t)); *amul_ms_out = t; }
    return MI_OK;
}

// ---------------------------------------------------------------------------
// host-only layout inspection (no device needed): lets tests verify the tiling
// on a CPU-only box by interpreting the tables themselves.  Not a compute path.
// ---------------------------------------------------------------------------
extern "C" int mi_layout_build_host(int32_t n_cells, int32_t n_faces, const int32_t* lower, const int32_t* upper,
                                    int32_t n_patches, const int32_t* patch_sizes, const int32_t* const* patch_face_cells,
                                    const int32_t* const* patch_nbr_cells, int32_t tile_cells, int32_t slot_cap, void** out)
{
    if (!out) return fail(MI_ERR_ARG, "mi_layout_build_host: out is NULL");
    TileLayout* L = new TileLayout();
    TileParams prm;
    if (tile_cells > 0) prm.tileCells = tile_cells;
    if (slot_cap > 0) prm.slotCap = slot_cap;
    prm.reorder = env_int("MI_TILE_REORDER", -1);
    const std::string err = build_tile_layout(n_cells, n_faces, lower, upper, n_patches, patch_sizes, patch_face_cells, prm, *L, patch_nbr_cells);
    if (!err.empty()) { delete L; return fail(MI_ERR_LIMIT, "mi_layout_build_host: " + err); }
    *out = L;
    return MI_OK;
}

// test hooks of the given-partition layout and of the inherited tiles (tiling.hpp: TileParams::givenPart, inherit_tiles)
extern "C" int mi_layout_build_host_given(int32_t n_cells, int32_t n_faces, const int32_t* lower, const int32_t* upper, int32_t n_parts, const int32_t* part, void** out)
{
    if (!out || !part) return fail(MI_ERR_ARG, "mi_layout_build_host_given: bad argument");
    TileLayout* L = new TileLayout();
    TileParams prm;
    prm.givenPart = part; prm.nGivenParts = n_parts;
    const std::string err = build_tile_layout(n_cells, n_faces, lower, upper, 0, nullptr, nullptr, prm, *L, nullptr);
    if (!err.empty()) { delete L; return fail(MI_ERR_LIMIT, "mi_layout_build_host_given: " + err); }
    *out = L;
    return MI_OK;
}
extern "C" int mi_layout_inherit_tiles(int32_t n_fine, const int32_t* restrict_map, const int32_t* fine_tile_of_cell, int32_t n_fine_tiles, int32_t n_coarse,
                                       int32_t n_coarse_faces, const int32_t* c_lower, const int32_t* c_upper, int32_t cell_cap, int32_t slot_cap,
                                       int32_t* part_out, int32_t* n_parts_out)
{
    if (!restrict_map || !fine_tile_of_cell || !part_out || !n_parts_out) return fail(MI_ERR_ARG, "mi_layout_inherit_tiles: bad argument");
    Table<int32_t> part;
    int32_t nParts = 0;
    const std::string err = inherit_tiles(n_fine, restrict_map, fine_tile_of_cell, n_fine_tiles, n_coarse, n_coarse_faces, c_lower, c_upper, 0, nullptr, nullptr,
                                          cell_cap > 0 ? cell_cap : 1024, slot_cap > 0 ? slot_cap : 4094, part, nParts);
    if (!err.empty()) return fail(MI_ERR_LIMIT, "mi_layout_inherit_tiles: " + err);
    std::copy(part.begin(), part.end(), part_out);
    *n_parts_out = nParts;
    return MI_OK;
}

extern "C" int mi_layout_array(void* handle, const char* name, const void** data, int64_t* len)
{
    if (!handle || !name || !data || !len) return fail(MI_ERR_ARG, "mi_layout_array: bad argument");
    TileLayout* L = static_cast<TileLayout*>(handle);
    const std::string n(name);
#define ARR(field) if (n == #field) { *data = L->field.data(); *len = (int64_t)L->field.size(); return MI_OK; }
    ARR(e2c) ARR(c2e) ARR(tileCellStart) ARR(tileSlotStart) ARR(tileIfaceSlot0) ARR(tileHaloStart) ARR(haloCell) ARR(tileSliceStart)
    ARR(sliceEntryStart) ARR(entries) ARR(sliceEntryStart16) ARR(entries16) ARR(slotBase) ARR(tileSbStart) ARR(slotFace) ARR(extSlot) ARR(interiorTiles) ARR(boundaryTiles)
    ARR(patchOffset) ARR(patchFaceCellsE) ARR(faceSlot)
#undef ARR
    return fail(MI_ERR_ARG, "mi_layout_array: unknown array " + n);
}

extern "C" int mi_layout_free(void* handle) { delete static_cast<TileLayout*>(handle); return MI_OK; }

// HIP-event pair helpers for callers that time a kernel range on the engine's stream
// (bench.py brackets the Amul phases of the distributed path).
extern "C" int mi_event_record(mi_matrix_t m, int32_t idx)
{
    if (!m || idx < 0) return fail(MI_ERR_ARG, "mi_event_record: bad argument");
    HIPCHK(hipSetDevice(m->addr->ctx->device));
    while (m->evPool.size() <= (size_t)idx) { hipEvent_t ev; HIPCHK(hipEventCreate(&ev)); m->evPool.push_back(ev); }
    HIPCHK(hipEventRecord(m->evPool[(size_t)idx], m->addr->ctx->stream));
    return MI_OK;
}
extern "C" int mi_event_elapsed_ms(mi_matrix_t m, int32_t idx0, int32_t idx1, float* ms)
{
    if (!m || !ms || idx0 < 0 || idx1 < 0 || (size_t)idx0 >= m->evPool.size() || (size_t)idx1 >= m->evPool.size()) return fail(MI_ERR_ARG, "mi_event_elapsed_ms: bad argument");
    HIPCHK(hipEventSynchronize(m->evPool[(size_t)idx1]));
    HIPCHK(hipEventElapsedTime(ms, m->evPool[(size_t)idx0], m->evPool[(size_t)idx1]));
    return MI_OK;
}

#include "multi_pipe.inc"
#include "multi.inc"
#include "comm.inc"
#include "persist.inc"
#include "pcg_fused.inc"
#include "gamg_engine.inc"
#include "assembly.inc"

mi_matrix_s::~mi_matrix_s()
{
    if (pcgGraph) (void)hipGraphExecDestroy(pcgGraph);
    if (mhostState) (void)hipHostFree(mhostState);
    for (auto* w : work) delete w;
    for (auto e : evPool) (void)hipEventDestroy(e);
    if (dpc) {
        if (dpc->commStream) (void)hipStreamDestroy(dpc->commStream);
        if (dpc->evPack) (void)hipEventDestroy(dpc->evPack);
        if (dpc->evHalo) (void)hipEventDestroy(dpc->evHalo);
        delete dpc->ph;
        delete dpc;
    }
}
