// nl_engine.hip -- host side of libnanollama_hip.so: weight upload / re-pack,
// state + KV-cache allocation, the per-token launch plan (hipGraph), and the
// extern "C" entry points declared in include/nanollama_hip.h.
//
// Reference being replaced (ariannamethod/nanollama): LoadLlamaModel
// go/model.go:121-174, loadWeights :177-265, allocState :324-343,
// precomputeRoPE :346-358, Forward :490-620, Reset :623-631, argmax
// go/main.go:400-408.  There is no CPU fallback in this library: without a
// HIP device every entry point that computes returns NL_ERR_HIP.
#include "../../include/nanollama_hip.h"
#include <mutex>
#include "nl_kernels.h"
#include "nl_qgemm.h"
#include "nl_qgemm2.h"
#include "nl_dgemm.h"
#include "nl_batch.h"
#include "nl_sample.h"
#include "nl_p2p.h"
#include "nl_block.h"
#include "nl_group.h"
#include "nl_tp.h"
#include "nl_persist.h"

#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace { void samp_free(nl::SampScratch &s); }

using namespace nl;

namespace {

thread_local std::string g_create_error;
// Allocation / graph-capture sections of different rank engines of ONE process (nl_create_group: a host thread per rank) must not
// overlap: a synchronous legacy-stream call of one thread (hipMalloc, hipMemset) fails while another thread's capture is open.
// Neither kind of section waits for a peer, so holding this across them cannot deadlock.
std::mutex g_setup_mu;

enum Kind { K_EMBED = 0, K_QKV, K_ATTN, K_WO, K_GATEUP, K_DOWN, K_LMHEAD, K_ARGMAX, K_ALLREDUCE, K_ATTNBLOCK, K_FFNBLOCK };
const char *kKindNames[NL_NUM_KINDS] = {"embed", "qkv_rope", "attention", "wo_resid", "gate_up_swiglu",
                                        "down_resid", "lm_head", "argmax", "allreduce", "attn_block", "ffn_block"};

struct PackedMat {
    uint8_t *q = nullptr;
    uint32_t *s = nullptr;
    uint8_t *q2 = nullptr;       // Q4_0 matrices the wide fused launches multiply on the matrix pipe: the same bytes with the chunks of
    uint32_t *s2 = nullptr;      // every (tile, 256-column group) permuted so that a quad of lanes holds four ROWS of a block (nl_tp.h)
    uint8_t *q3 = nullptr;       // Q4_0 layer matrices of a model whose short multi-token steps run on dgemm_kernel: the block-major copy
    uint32_t *s3 = nullptr;      // (nl_dgemm.h dg_permute_kernel), built at the first such step
    int wtype = -1, src_type = -1, rows = 0, cols = 0, ntiles = 0, npairs = 0;  // wtype = device layout type
    size_t q_bytes = 0, s_bytes = 0;
    bool ready = false;
};

// device layout type of a source ggml type (Q5_0 is expanded into the Q8_0 layout by repack_kernel)
int device_type(int src) { return src == WT_Q5_0 ? WT_Q8_0 : src; }
int chunks_per_pair(int wt) { return (wt == WT_Q8_0 || wt == WT_Q6_K) ? 4 : (wt == WT_Q4_0 || wt == WT_Q4_K) ? 2 : wt == WT_F16 ? 8 : 16; }
bool is_scaled(int wt) { return wt == WT_Q8_0 || wt == WT_Q4_0 || wt == WT_Q4_K || wt == WT_Q6_K; }
bool type_supported(uint32_t t) {
    return t == WT_F32 || t == WT_F16 || t == WT_Q4_0 || t == WT_Q5_0 || t == WT_Q8_0 || t == WT_Q4_K || t == WT_Q6_K;
}
bool is_kquant(uint32_t t) { return t == WT_Q4_K || t == WT_Q6_K; }
size_t raw_bytes(uint32_t t, uint64_t nel) {
    switch (t) {
    case WT_F32: return nel * 4;
    case WT_F16: return nel * 2;
    case WT_Q4_0: return nel / 32 * 18;
    case WT_Q5_0: return nel / 32 * 22;
    case WT_Q8_0: return nel / 32 * 34;
    case WT_Q4_K: return nel / 256 * 144;
    case WT_Q6_K: return nel / 256 * 210;
    default: return 0;
    }
}

// --- RCCL, bound lazily so single-GPU use never loads it ---------------------
struct NcclId { char internal[128]; };
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(void **, int, NcclId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load(std::string &err) {
        if (lib) return true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("dlopen librccl failed: ") + dlerror(); return false; }
        GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllReduce || !AllGather || !CommDestroy) {
            err = "librccl is missing required symbols";
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
constexpr int kNcclFloat = 7, kNcclSum = 0;

struct Op {
    int kind;
    int coll;  // 0 none, 1 all-reduce(sum) of buf[count], 2 all-gather into buf (count per rank),
               // 3 (in-process group only) all-reduce(sum) of buf[count], then add_to[i] += buf[i]
    float *buf;
    size_t count;
    std::function<hipError_t(hipStream_t)> fn;
    float *add_to = nullptr;
    // GEMV launches keep their parameters here so a later pass can point each one at its successor
    bool is_gemv = false;
    int wtype = 0, pro = 0, epi = 0;
    GemvParams gp{};
};

}  // namespace

struct nl_engine {
    nl_config cfg{};
    std::string err;
    int dev = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool finalized = false;

    // shard dimensions (== full dimensions when tp_size == 1)
    int G = 1, rank = 0;
    int Hs = 0, KVs = 0, Is = 0, Vs = 0, hd = 0, gqa = 1;
    int nsplit_max = 1;

    struct Layer {
        PackedMat qkv, wo, gate, up, down;
        PackedMat wo_head;   // per-head 64-column slices of WO for the fused attention block (nl_block.h); small models only
        PackedMat dn_slice;  // W_down sliced by 256 columns for the fused feed-forward block (nl_block.h)
        float *attn_norm = nullptr, *ffn_norm = nullptr;
        float *bq = nullptr, *bk = nullptr, *bv = nullptr, *bo = nullptr;  // optional biases (this rank's slice)
        bool have_q = false, have_k = false, have_v = false;
    };
    std::vector<Layer> layers;
    PackedMat lm_head;
    uint8_t *embd_raw = nullptr;
    int embd_type = -1;
    size_t embd_bytes = 0;
    bool have_output = false;
    float *output_norm = nullptr;

    int *gamma_row = nullptr;      // [vocab] -> row of gamma_val or -1 (go/gamma.go IndexMap)
    float *gamma_val = nullptr;    // [n][dim]
    float *rope_cos = nullptr, *rope_sin = nullptr;
    float *x[2] = {nullptr, nullptr};
    float *qbuf = nullptr, *part_o = nullptr, *part_ml = nullptr, *hb = nullptr, *ar = nullptr, *logits = nullptr;
    float *kcache = nullptr, *vcache = nullptr;
    long long kv_layer_stride = 0, kv_stream_stride = 0;
    int *ctl = nullptr, *ids = nullptr, *result = nullptr;
    float *amax_val = nullptr;
    int *amax_idx = nullptr;
    int amax_slots = 0;
    int *h_ctl = nullptr;  // pinned staging
    // Per-call Forward on the launch plans (go/main.go:173-219 calls Forward once per token): the caller's token / position travel
    // through a pinned, device-visible box the first launch of the step's graph reads, completion and the argmax come back through
    // a pinned done word the host spins on -- no control copy, no stream synchronise per token (one GPU, no collectives)
    int *h_box = nullptr, *d_box = nullptr;              // [0, CTL_WORDS): control words, [8]: sequence number of the call
    unsigned long long *h_done = nullptr, *d_done = nullptr;
    unsigned box_seq = 0;
    float *d_h_logits = nullptr; // device address of h_logits (the LM head of a per-call Forward stores the logits there itself)
    float *h_logits = nullptr;   // pinned [vocab] + one int behind it: the per-call read-backs of nl_forward / nl_forward_argmax /
                                 // nl_prefill land here by DMA and are copied to the caller's (pageable) buffer by the CPU -- a
                                 // hipMemcpyAsync to pageable memory is a staged, blocking copy inside the runtime (128 KB: ~90 us)
    std::vector<int> hw;   // per stream: positions [0, hw) hold K/V written since the last nl_reset
    int *h_ctl_ring = nullptr;  // pinned: one ctl block per queued step (prefill / batch)
    int ctl_ring_cap = 0;
    int ids_cap = 0;
    size_t bytes_weights = 0, bytes_kv = 0, bytes_state = 0;

    // weights live in a few large allocations (2 MiB-friendly) instead of one hipMalloc per tensor
    std::vector<void *> arena_chunks;
    char *arena_cur = nullptr;
    size_t arena_left = 0;
    uint8_t *stage = nullptr;  // grow-only device staging buffer for raw GGUF bytes
    size_t stage_cap = 0;

    // multi-token step (batched decode streams / prefill): MFMA path, allocated on first use
    struct Batch {
        bool ready = false;
        int cap = 0, lm_cap = 0;
        uint4 *xfrag = nullptr;  // fp16 hi/lo activation fragments of the GEMM being run
        uint4 *xfrag2 = nullptr; // second store: output of the fused gate/up/SwiGLU GEMM, input of down
        double *ssq = nullptr;   // [cap][dim / 64] sums of squares of the residual rows (RMSNorm folded around the prompt GEMMs)
        float *nscale = nullptr; // [2][cap] per-token power-of-two pre-scales of the folded norm's fragments (norm_prescale, nl_qgemm.h)
        float *kpart = nullptr, *kpart2 = nullptr;  // split-K partial sums (second buffer: up, alive beside gate)
        size_t kpart_cap = 0;    // floats, each
        float *x = nullptr, *qkv = nullptr, *q = nullptr, *g = nullptr, *u = nullptr, *logits = nullptr,
              *part_o = nullptr, *part_ml = nullptr;
        int *tok = nullptr, *pos = nullptr, *stream = nullptr, *ids = nullptr;
        float *tcos = nullptr, *tsin = nullptr;      // [cap][hd / 2]: the RoPE rows of every token's position (bembed_kernel; QGemmParams::Rope::tcos)
        long long *tkv = nullptr;                    // [cap]: stream * kv_stream_stride + pos * hd
        int *h_meta = nullptr;  // pinned [5][cap]: token | position | stream | attention workgroup list | partials per token
        uint4 *kv16 = nullptr;  // one layer's K / V^T of a prompt's stream as fp16 hi / lo LDS images (nl_batch.h Kv16Image)
    } bt;

    // Concurrent sub-batches of a decode batch (nl_forward_batch): the streams of a batch are independent (go/serve.go:106-108
    // serialises them only because it has one CPU engine), so the batch is cut into groups that step on their own HIP stream
    // with their own step buffers -- one dependent chain's launch heads and memory round trips overlap another's work.
    // Each group's step is a hipGraph cached by (tokens, position splits): twice the launches would otherwise bind on the host.
    struct SubBatch {
        Batch bt;
        hipStream_t st = nullptr;
        hipEvent_t done = nullptr;
        struct G { int n, nsplit; hipGraph_t graph; hipGraphExec_t exec; };
        std::vector<G> graphs;
    } sub[4];
    int dg_state = 0;             // dgemm_kernel's weight copies: 0 not looked at yet, 1 built, -1 the model does not qualify
    bool dg_head = false;         // ... and the LM head has one too (decode batches: dgemm_kernel with the argmax candidates in its epilogue)
    std::vector<SubBatch::G> bt_graphs;   // step graphs of the whole-batch decode step (bt), same key
    hipEvent_t sub_fork = nullptr;
    int sub_batches = 1;          // NL_SUB_BATCHES: groups a decode batch is cut into (1 = the whole batch as one step: measured fastest, profiles/r04_subbatch_groups.log)

    // on-device sampling (nl_sample_decode): scratch for one vocabulary, allocated on first use
    SampScratch sp;
    bool sp_ready = false;
    float *samp_keep = nullptr;   // logits as nl_sample_decode found them (restart point of a redo on the general plan)
    int sp_uniforms_cap = 0;

    // A launch plan with its captured graphs.  ps[0]: five launches per layer, any context length.  ps[1] (small
    // models, nl_block.h): the attention half of a layer as one launch -- chosen per step while the context is short.
    struct PlanSet {
        std::vector<Op> ops;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipGraph_t multi = nullptr;          // the same plan graph_steps times: chained greedy decode replays it
        hipGraphExec_t multi_exec = nullptr;  // (one inter-graph gap per graph_steps tokens instead of per token)
        hipGraph_t call = nullptr;           // the plan between a fetch of the step's control words from the pinned call box and a
        hipGraphExec_t call_exec = nullptr;   // store of {sequence, argmax} into its done word: one nl_forward, no copy, no synchronise
    } ps[3];                                 // (ps[2]: ps[1] with the attention launch whose passes are shared by helper blocks: wide tier, one GPU, from the second pass on)
    bool fused = false;           // ps[1] exists
    int fused_mode = 0;           // 1: whole attention half per layer (nl_block.h); 2: projection + attention (nl_group.h);
                                  // 3: a tensor-parallel rank's layer as two launches (nl_tp.h); 4: one GPU, wide tier: mode 3's
                                  // attention half with a direct seam (projection + attention + WO), then the two GEMVs
    bool mf_attn = false, mf_ffn = false;   // the attention / feed-forward launch of modes 3 / 4 multiplies on the matrix pipe (NL_MFMA_DOT=0: off)
    int prefetch = 1;             // NL_PREFETCH bits (nl_tp.h PfTiles): 1 = the attention launch warms round 0 of the feed-forward launch (kept:
                                  // big 670 -> 698 tok/s); measured and left off (profiles/r05_prefetch_ab.log): 2 = the feed-forward launch warms
                                  // the next layer's projection tiles (+0.8 %), 4 = a feed-forward round warms the round after the next (-5 %: two
                                  // rounds and the touched one exceed the L2), 8 = bit 1's touches issued right after the block's own projection
    bool wide_ffn = false;        // modes 3 / 4: the feed-forward half is wide_ffn_kernel (nl_tp.h): always in mode 4 when eligible, in
                                  // mode 3 when the rank's shard is too large for tp_ffn_kernel's one-tile producers (tp 2 of the 7.9B tier)
    int wide_nf = 1, wide_ngc = 1;
    struct TpGeom {               // mode 3 geometry, fixed at nl_finalize
        int wo_gshift = 0;        // log2 of the 256-column groups of a WO row
        int wo_tpw = 1;           // WO tiles per block of the attention launch's grid
        int pair = 0, n_prod = 0, n_cons = 0, ct_shift = 1, grid = 0, nf = 1, ngc = 1, nr = 1;   // feed-forward launch
    } tpg;
    u32x4 *tp_xq = nullptr;       // 16-byte granules of the two-launch layer (nl_tp.h): q | k | v tiles of a kv group,
    u32x4 *tp_xo = nullptr;       //   the heads' attention outputs,
    u32x4 *tp_xp = nullptr;       //   the helper blocks' pass records (long contexts on one GPU),
    bool attn_helpers = false;    //   ... when the launch's geometry has three helpers for every head (NL_ATTN_HELPERS=0: off)
    u32x4 *tp_hx = nullptr;       //   g | u (or h) tiles of the feed-forward half
    int grp_tpm = 1;              // mode 2: 16-row tiles per workgroup
    bool grp_grid_fits = false;   // modes 3 / 4: the projection grid (incl. its blocks without a tile, which stay) fits the compute units
    bool ffn_fused = false;       // mode 1: the feed-forward half is one launch too (ffn_block_kernel)
    float *parts_ffn = nullptr;   // [I / 256][D] per-slice W_down partials
    unsigned long long *xchg_ffn = nullptr;   // [I] granules
    int fused_max_pos = 0;        // ps[1] serves steps whose position is below this
    float *parts = nullptr;       // [Hs][D] per-head WO partials of the fused block
    unsigned long long *xchg = nullptr;   // [Hs][192] granules exchanged inside a head's cluster
    unsigned *tick = nullptr;     // {forward counter, status} device words of the fused block
    long long *dbg_block = nullptr;
    unsigned *h_status = nullptr; // pinned, device-visible: non-zero after an in-kernel exchange gave up
    int spin_limit = 400000;      // polls before a cluster exchange gives up (NL_FUSED_SPIN_LIMIT)
    int num_cus = 256;            // compute units of the device: a fused launch must fit on them all at once
    bool fused_retired = false;   // an exchange timed out once: the handle stays on the general plan
    int graph_steps = 1;
    EmbedParams plan_embed{};    // kept for the fused argmax + embed launch of the multi-step graph
    ArgmaxParams plan_argmax{};
    P2PArgmaxParams plan_p2p_argmax{};
    // sampled chained decode: {sampler, plan} x graph_steps, captured per sampling-parameter set
    hipGraph_t samp_graph[3] = {nullptr, nullptr, nullptr};          // per launch plan (ps[0] / ps[1] / ps[2])
    hipGraphExec_t samp_graph_exec[3] = {nullptr, nullptr, nullptr};
    nl_sample_params samp_graph_params[3]{};
    bool samp_graph_failed = false;
    bool use_graph = true;
    void *comm = nullptr;
    // one-shot push all-reduce between the ranks of a node (nl_p2p.h); replaces RCCL on the data path when set up
    struct P2P {
        bool on = false;          // nl_p2p_import succeeded: the plan uses the push seams
        void *area = nullptr;     // this rank's receive area (uncached, exported over hipIpc)
        size_t bytes = 0;
        bool uncached = false;
        bool loopback = false;    // nl_p2p_loopback: every "peer" is this rank's own area (one rank alone, timing only)
        void *peer[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // rank r's area mapped here
        bool opened[8] = {false, false, false, false, false, false, false, false};
        unsigned *epoch = nullptr, *status = nullptr;   // device words: forward counter, poll-timeout flags
        long long timeout_ticks = 0;
        size_t off_amax = 0, off_logits = 0;             // byte offsets inside an area
    } p2p;
    // weight-stationary, XCD-pipelined persistent greedy decode of the smallest tier (nl_persist.h): one launch decodes a
    // whole chunk of tokens with every layer's weights resident in the registers of one XCD's compute units
    struct Persist {
        bool candidate = false;      // shape / type admit the path; raw Q8_0 tensors are kept on the device until nl_finalize
        bool ready = false;          // images built: nl_decode_greedy takes it for chunks that end below max_pos
        bool retired = false;        // a poll gave up once (or the census was not 8 x 32 three times): the handle keeps the launch plans
        int fake_misses = 0;         // (NL_PERSIST_FAKE_CENSUS_MISS: misses reported so far)
        int census_misses = 0;       // launches that found their workgroups placed otherwise (a transient: another launch held compute units)
        uint8_t *raw[PD_MAXL][7] = {};
        uint8_t *lm_raw = nullptr;
        unsigned char rtype[PD_MAXL][7] = {}, lm_type = 0;     // block type of every kept tensor (Q8_0 / Q4_0 / Q5_0)
        uint8_t *embd_q8 = nullptr;  // token_embd of a Q4_0 / Q5_0 file re-blocked as Q8_0 for the in-launch lookup
        uint4 *wimg = nullptr, *lmimg = nullptr;
        unsigned short *simg = nullptr, *lmsimg = nullptr;
        pd_u64 *gx = nullptr, *gqkv = nullptr, *go = nullptr, *gxp = nullptr, *gh = nullptr;
        pd_u64 *gpart = nullptr;
        pd_u32x4 *gam = nullptr;
        unsigned *census = nullptr, *status = nullptr, *h_status = nullptr;
        float *norms = nullptr;      // [L][2][D] + [D]: attn_norm | ffn_norm of every layer, output_norm (one pointer for the kernel)
        unsigned tag_base = 0;
        int max_pos = PD_MAX_POS;
        int spin_limit = 2000000;
        long long *dbg = nullptr;
        long long launches = 0, tokens = 0;
        // resident session (per-call Forward): the launch stays on the chip between nl_forward calls and takes every next
        // token from a pinned mailbox word; see PdParams::session
        bool session_on = true;      // NL_PERSIST_SESSION=0: one launch of one step per call instead
        bool live = false;           // a session launch is (or may still be) resident on e->stream
        bool s_logits = false;       // ... whose LM-head units store every step's logits into h_logits
        int s_stream = 0, s_next_pos = 0, s_step = 0, s_nsteps = 0;
        long long idle_ticks = 200000;   // 2 ms of the 100 MHz clock (NL_PERSIST_IDLE_US)
        pd_u64 *h_mail = nullptr, *d_mail = nullptr;   // pinned: [0] mailbox, [8] done word (own cache lines); device view
        pd_u64 *gtok = nullptr;
        unsigned launch_no = 0;
        long long sessions = 0;
    } pd;
    // One-process tensor-parallel group (nl_create_group): THIS handle is the leader the caller holds -- it owns no device
    // state -- and grp->members are the rank engines (one per device, tp_rank = index), wired through the push all-reduce
    // with plain peer pointers.  Every entry point called on the leader runs the same entry point on every member, each
    // on its own host thread (a rank's step cannot finish before its peers have launched theirs).
    struct GroupCtl *grp = nullptr;
    int tw_override = 0, kw_override = 0;
    bool force_tp_plan = false;  // NL_FORCE_TP_PLAN: use the all-reduce / all-gather seams even with one rank

    int fail(int code, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define HIPCK(e, expr)                                                                               \
    do {                                                                                             \
        hipError_t _s = (expr);                                                                      \
        if (_s != hipSuccess) return (e)->fail(NL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_s)); \
    } while (0)

namespace {

// bytes of one packed 16-row tile of a matrix with `npairs` 64-column pairs per row (nl_kernels.h load_pair)
inline unsigned tile_qbytes(int wt, int npairs) {
    const int cpp = wt == WT_Q8_0 ? 4 : wt == WT_Q4_0 ? 2 : wt == WT_F16 ? 8 : wt == WT_F32 ? 16 : wt == WT_Q4_K ? 2 : 4;
    return (unsigned)npairs * (unsigned)cpp * TR * 16u;
}
inline unsigned tile_sbytes(int wt, int npairs) { return (unsigned)npairs * TR * 4u * (unsigned)scale_words(wt); }

template <typename T>
hipError_t dalloc(T **p, size_t n, size_t *acct = nullptr) {
    hipError_t s = hipMalloc((void **)p, n * sizeof(T) ? n * sizeof(T) : sizeof(T));
    if (s == hipSuccess && acct) *acct += n * sizeof(T);
    return s;
}

hipError_t arena_alloc(nl_engine *e, void **out, size_t bytes) {
    const size_t kChunk = (size_t)256 << 20;
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > e->arena_left) {
        size_t sz = std::max(bytes, kChunk);
        void *c = nullptr;
        hipError_t s = hipMalloc(&c, sz);
        if (s != hipSuccess) return s;
        e->arena_chunks.push_back(c);
        e->arena_cur = (char *)c;
        e->arena_left = sz;
    }
    *out = e->arena_cur;
    e->arena_cur += bytes;
    e->arena_left -= bytes;
    return hipSuccess;
}

hipError_t stage_reserve(nl_engine *e, size_t bytes) {
    if (bytes <= e->stage_cap) return hipSuccess;
    if (e->stage) hipFree(e->stage);
    e->stage = nullptr;
    e->stage_cap = 0;
    size_t cap = std::max(bytes, (size_t)64 << 20);
    hipError_t s = hipMalloc((void **)&e->stage, cap);
    if (s == hipSuccess) e->stage_cap = cap;
    return s;
}

// Re-pack a slice of a raw GGUF tensor (already on the device) into tiles
// [tile0, tile0 + ntiles) of a PackedMat.
hipError_t repack(nl_engine *e, PackedMat &m, const uint8_t *d_src, int wtype, int src_cols, int row0, int nrows,
                  int col0, int ncols, int tile0, int ntiles, int rowmap) {
    const int cpp = chunks_per_pair(device_type(wtype));
    RepackParams P{};
    P.src = d_src;
    P.q = m.q + (size_t)tile0 * m.npairs * cpp * TR * 16;
    P.s = m.s ? m.s + (size_t)tile0 * m.npairs * TR * scale_words(wtype) : nullptr;
    P.wtype = wtype;
    P.src_cols = src_cols;
    P.row0 = row0; P.nrows = nrows; P.col0 = col0; P.ncols = ncols;
    P.ntiles = ntiles; P.npairs = m.npairs;
    P.rowmap = rowmap; P.head_dim = e->hd;
    long long nchunks = (long long)ntiles * m.npairs * cpp * TR;
    int blocks = (int)std::min<long long>((nchunks + 255) / 256, 8192);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(repack_kernel, dim3(blocks), dim3(256), 0, e->stream, P);
    return hipGetLastError();
}

hipError_t alloc_packed(nl_engine *e, PackedMat &m, int wtype, int rows_padded_tiles, int rows, int cols) {
    m.src_type = wtype;
    m.wtype = wtype = device_type(wtype);
    m.rows = rows;
    m.cols = cols;
    m.ntiles = rows_padded_tiles;
    m.npairs = (cols + PAIR - 1) / PAIR;
    m.q_bytes = (size_t)m.ntiles * m.npairs * chunks_per_pair(wtype) * TR * 16;
    m.s_bytes = is_scaled(wtype) ? (size_t)m.ntiles * m.npairs * TR * 4 * scale_words(wtype) : 0;
    hipError_t s = arena_alloc(e, (void **)&m.q, m.q_bytes);
    if (s != hipSuccess) return s;
    if (m.s_bytes) {
        s = arena_alloc(e, (void **)&m.s, m.s_bytes);
        if (s != hipSuccess) return s;
    }
    e->bytes_weights += m.q_bytes + m.s_bytes;
    return hipSuccess;
}

void choose_geometry(const nl_engine *e, const PackedMat &m, int &tw, int &kw, int mats = 1) {
    // kw wavefronts share a tile's 256-column groups, tw tiles share a workgroup.  Aim for >= ~2048
    // wavefronts in flight (8 per CU); small matrices go all the way to one group per wavefront,
    // which is what minimises the in-launch latency of small models.
    const int ngroups = (m.npairs + KL - 1) / KL;
    static const int target = getenv("NL_WAVES") ? atoi(getenv("NL_WAVES")) : 2048;   // developer knob (tools/)
    kw = (target + m.ntiles - 1) / std::max(1, m.ntiles);
    const int cap = 8 / mats;  // workgroups are capped at 8 wavefronts (256 VGPRs each)
    kw = (kw + mats - 1) / mats;
    kw = std::max(1, std::min(kw, std::min(ngroups, cap)));
    tw = std::max(1, 4 / (kw * mats));
    if (e->kw_override > 0) kw = std::min(e->kw_override, cap);
    if (e->tw_override > 0) tw = e->tw_override;
    if (tw * kw * mats > 8) tw = std::max(1, 8 / (kw * mats));
    if (tw > m.ntiles) tw = m.ntiles;
}

template <int PRO, int EPI, int NP = 1>
hipError_t launch_gemv_t(int wtype, GemvParams P, hipStream_t st) {
    P.add_src = P.add ? P.add : P.x;   // PRO_ATTN kernels never read it
    P.kw_inv = udiv_inv((unsigned)P.kw); P.kw2_inv = udiv_inv(2u * (unsigned)P.kw);
    const int nwaves = P.tw * P.kw * (EPI == EPI_SWIGLU ? 2 : 1);
    const size_t lds = (size_t)nwaves * XS_WAVE * 4 + (size_t)nwaves * TR * 4 + (size_t)nwaves * 8;
    const dim3 grid((P.ntiles + P.tw - 1) / P.tw), block(nwaves * 64);
    if constexpr (PRO == PRO_NORM_PARTS) {   // consumer of the fused attention block: Q8_0 / Q4_0 models only
        switch (wtype) {
        case WT_Q8_0: hipLaunchKernelGGL((gemv_kernel<WT_Q8_0, PRO, EPI, NP>), grid, block, lds, st, P); break;
        case WT_Q4_0: hipLaunchKernelGGL((gemv_kernel<WT_Q4_0, PRO, EPI, NP>), grid, block, lds, st, P); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    } else {
    switch (wtype) {
    case WT_Q8_0: hipLaunchKernelGGL((gemv_kernel<WT_Q8_0, PRO, EPI>), grid, block, lds, st, P); break;
    case WT_Q4_0: hipLaunchKernelGGL((gemv_kernel<WT_Q4_0, PRO, EPI>), grid, block, lds, st, P); break;
    case WT_F16: hipLaunchKernelGGL((gemv_kernel<WT_F16, PRO, EPI>), grid, block, lds, st, P); break;
    case WT_F32: hipLaunchKernelGGL((gemv_kernel<WT_F32, PRO, EPI>), grid, block, lds, st, P); break;
    case WT_Q4_K: hipLaunchKernelGGL((gemv_kernel<WT_Q4_K, PRO, EPI>), grid, block, lds, st, P); break;
    case WT_Q6_K: hipLaunchKernelGGL((gemv_kernel<WT_Q6_K, PRO, EPI>), grid, block, lds, st, P); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
    }
}

GemvParams base_params(const nl_engine *e, const PackedMat &m, int mats = 1) {
    GemvParams P{};
    P.q0 = m.q; P.s0 = m.s;
    P.rows = m.rows; P.cols = m.cols; P.npairs = m.npairs; P.ntiles = m.ntiles;
    choose_geometry(e, m, P.tw, P.kw, mats);
    P.eps = e->cfg.rms_eps;
    P.ctl = e->ctl;
    P.head_dim = e->hd;
    P.nsplit_max = e->nsplit_max;
    return P;
}

template <int HD>
hipError_t launch_attn_hd(int gqa, AttnParams P, dim3 grid, hipStream_t st) {
    switch (gqa) {
    case 1: hipLaunchKernelGGL((attn_kernel<HD, 1>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 2: hipLaunchKernelGGL((attn_kernel<HD, 2>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 3: hipLaunchKernelGGL((attn_kernel<HD, 3>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 4: hipLaunchKernelGGL((attn_kernel<HD, 4>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 8: hipLaunchKernelGGL((attn_kernel<HD, 8>), grid, dim3(ATT_THREADS), 0, st, P); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int HD>
hipError_t launch_attn_fin_hd(int gqa, AttnParams P, dim3 grid, hipStream_t st) {
    switch (gqa) {
    case 1: hipLaunchKernelGGL((attn_kernel<HD, 1, true>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 2: hipLaunchKernelGGL((attn_kernel<HD, 2, true>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 3: hipLaunchKernelGGL((attn_kernel<HD, 3, true>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 4: hipLaunchKernelGGL((attn_kernel<HD, 4, true>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 8: hipLaunchKernelGGL((attn_kernel<HD, 8, true>), grid, dim3(ATT_THREADS), 0, st, P); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// several position splits per token with the RoPE prologue (decode batches at positions >= 128, AttnParams::rp)
template <int HD>
hipError_t launch_attn_rope_hd(int gqa, AttnParams P, dim3 grid, hipStream_t st) {
    switch (gqa) {
    case 1: hipLaunchKernelGGL((attn_kernel<HD, 1, false, true>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 2: hipLaunchKernelGGL((attn_kernel<HD, 2, false, true>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 3: hipLaunchKernelGGL((attn_kernel<HD, 3, false, true>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 4: hipLaunchKernelGGL((attn_kernel<HD, 4, false, true>), grid, dim3(ATT_THREADS), 0, st, P); break;
    case 8: hipLaunchKernelGGL((attn_kernel<HD, 8, false, true>), grid, dim3(ATT_THREADS), 0, st, P); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_attn(int hd, int gqa, AttnParams P, dim3 grid, hipStream_t st) {
    if (P.rp.on) return hd == 64 ? launch_attn_rope_hd<64>(gqa, P, grid, st) : hd == 32 ? launch_attn_rope_hd<32>(gqa, P, grid, st) : hipErrorInvalidValue;
    if (hd == 64) return launch_attn_hd<64>(gqa, P, grid, st);
    if (hd == 32) return launch_attn_hd<32>(gqa, P, grid, st);
    return hipErrorInvalidValue;
}

// developer knobs of the prompt attention, read ONCE per process and in one place: the workgroup list of batched_step
// and the launchers below must agree on which kernel runs (a list built for attn_tile16_kernel is not a grid of the f32
// tile kernel)
struct AttnKnobs { bool f32_tile, no_tile, no_fin; };
static const AttnKnobs &attn_knobs() {
    static const AttnKnobs k{getenv("NL_ATTN_F32") != nullptr,       // the f32-MFMA tile kernel
                             getenv("NL_NO_ATTN_TILE") != nullptr,   // no tile kernel at all: attn_kernel per token
                             getenv("NL_NO_ATTN_FIN") != nullptr};
    return k;
}

template <int HD, int G>
void launch_attn_tile_g(const AttnParams &P, int n, int kvs, int nsplit, hipStream_t st) {
    constexpr int QT = AttnTileQT<G>::value;
    const bool f32_tile = attn_knobs().f32_tile;
    if (f32_tile) hipLaunchKernelGGL((attn_tile_kernel<HD, G, QT>), dim3(kvs, nsplit, (n + QT - 1) / QT), dim3(QT * G * 4), 0, st, P, n);
    else {
        constexpr int QT16 = AttnTile16QT<G>::value;
        if (P.live_map && P.kv16) hipLaunchKernelGGL((attn_tile16_kernel<HD, G, AttnTile16ShadowQT<G>::value, true>), dim3(nsplit /* = listed workgroups */), dim3(AttnTile16ShadowQT<G>::value * G * 4), 0, st, P, n);
        else if (P.live_map) hipLaunchKernelGGL((attn_tile16_kernel<HD, G, QT16, false>), dim3(nsplit /* = listed workgroups */), dim3(QT16 * G * 4), 0, st, P, n);
        else hipLaunchKernelGGL((attn_tile16_kernel<HD, G, QT16, false>), dim3(kvs, nsplit, (n + QT16 - 1) / QT16), dim3(QT16 * G * 4), 0, st, P, n);
    }
}
inline int attn_tile16_qt(int gqa) { return NL_ATT16_MUL * (gqa == 1 ? 64 : gqa == 2 ? 32 : gqa == 8 ? 8 : 16); }   // = AttnTile16QT<G>::value
inline bool attn_tile_supported(int gqa) { return gqa == 1 || gqa == 2 || gqa == 3 || gqa == 4 || gqa == 8; }
template <int HD>
hipError_t launch_attn_tile_hd(int gqa, const AttnParams &P, int n, int kvs, int nsplit, hipStream_t st) {
    switch (gqa) {
    case 1: launch_attn_tile_g<HD, 1>(P, n, kvs, nsplit, st); break;
    case 2: launch_attn_tile_g<HD, 2>(P, n, kvs, nsplit, st); break;
    case 3: launch_attn_tile_g<HD, 3>(P, n, kvs, nsplit, st); break;
    case 4: launch_attn_tile_g<HD, 4>(P, n, kvs, nsplit, st); break;
    case 8: launch_attn_tile_g<HD, 8>(P, n, kvs, nsplit, st); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_gemv_dyn(int wt, int pro, int epi, const GemvParams &P, hipStream_t st) {
    if (pro == PRO_NORM && epi == EPI_QKV) return launch_gemv_t<PRO_NORM, EPI_QKV>(wt, P, st);
    if (pro == PRO_ATTN && epi == EPI_RESID) return launch_gemv_t<PRO_ATTN, EPI_RESID>(wt, P, st);
    if (pro == PRO_ATTN && epi == EPI_STORE) return launch_gemv_t<PRO_ATTN, EPI_STORE>(wt, P, st);
    if (pro == PRO_NORM && epi == EPI_SWIGLU) return launch_gemv_t<PRO_NORM, EPI_SWIGLU>(wt, P, st);
    if (pro == PRO_PLAIN && epi == EPI_RESID) return launch_gemv_t<PRO_PLAIN, EPI_RESID>(wt, P, st);
    if (pro == PRO_PLAIN && epi == EPI_STORE) return launch_gemv_t<PRO_PLAIN, EPI_STORE>(wt, P, st);
    if (pro == PRO_NORM && epi == EPI_STORE) return launch_gemv_t<PRO_NORM, EPI_STORE>(wt, P, st);
    if (pro == PRO_NORM_PARTS && epi == EPI_SWIGLU) {   // exactly as many part loads as heads, for the common head counts
        if (P.nparts <= 4) return launch_gemv_t<PRO_NORM_PARTS, EPI_SWIGLU, 4>(wt, P, st);
        if (P.nparts <= 9) return launch_gemv_t<PRO_NORM_PARTS, EPI_SWIGLU, 9>(wt, P, st);
        return launch_gemv_t<PRO_NORM_PARTS, EPI_SWIGLU, MAX_PARTS>(wt, P, st);
    }
    if (pro == PRO_ATTN && epi == EPI_P2P) return launch_gemv_t<PRO_ATTN, EPI_P2P>(wt, P, st);
    if (pro == PRO_PLAIN && epi == EPI_P2P) return launch_gemv_t<PRO_PLAIN, EPI_P2P>(wt, P, st);
    return hipErrorInvalidValue;
}

void push_gemv(std::vector<Op> &plan, int kind, int coll, float *buf, size_t count, int wt, int pro, int epi, const GemvParams &P) {
    Op op{kind, coll, buf, count, nullptr};
    op.is_gemv = true; op.wtype = wt; op.pro = pro; op.epi = epi; op.gp = P;
    plan.push_back(op);
}

// Freeze the launch closures once every GEMV's parameters are final.
void link_prefetch(std::vector<Op> &plan) {
    for (Op &cur : plan) {
        if (!cur.is_gemv) continue;
        const int wt = cur.wtype, pro = cur.pro, epi = cur.epi;
        const GemvParams P = cur.gp;
        cur.fn = [wt, pro, epi, P](hipStream_t st) { return launch_gemv_dyn(wt, pro, epi, P, st); };
    }
}

template <int WT>
hipError_t launch_attn_block(const BlockParams &B, int npin, int grid, size_t lds, hipStream_t st) {
    if (npin == 0) hipLaunchKernelGGL((attn_block_kernel<WT, 0>), dim3(grid), dim3(BLK_THREADS), lds, st, B);
    else if (npin <= 2) hipLaunchKernelGGL((attn_block_kernel<WT, 2>), dim3(grid), dim3(BLK_THREADS), lds, st, B);
    else if (npin <= 6) hipLaunchKernelGGL((attn_block_kernel<WT, 6>), dim3(grid), dim3(BLK_THREADS), lds, st, B);
    else hipLaunchKernelGGL((attn_block_kernel<WT, FFN_MAX_PARTS>), dim3(grid), dim3(BLK_THREADS), lds, st, B);
    return hipGetLastError();
}
template <int WT>
hipError_t launch_ffn_block(const FfnParams &F, int grid, hipStream_t st) {
    const size_t lds = ffn_lds_bytes();
    if (F.nparts_in <= 4) hipLaunchKernelGGL((ffn_block_kernel<WT, 4>), dim3(grid), dim3(FFN_THREADS), lds, st, F);
    else if (F.nparts_in <= 9) hipLaunchKernelGGL((ffn_block_kernel<WT, 9>), dim3(grid), dim3(FFN_THREADS), lds, st, F);
    else hipLaunchKernelGGL((ffn_block_kernel<WT, BLK_MAX_PARTS>), dim3(grid), dim3(FFN_THREADS), lds, st, F);
    return hipGetLastError();
}

BlockParams attn_block_params(nl_engine *e, int l, const float *x_in) {
    const nl_config &c = e->cfg;
    nl_engine::Layer &L = e->layers[l];
    BlockParams B{};
    B.qkv_q = L.qkv.q; B.qkv_s = L.qkv.s; B.wo_q = L.wo_head.q; B.wo_s = L.wo_head.s;
    B.D = c.dim; B.npairs = L.qkv.npairs; B.n_q_heads = e->Hs; B.n_kv_heads = e->KVs; B.seq_len = c.seq_len;
    B.rope_conj = c.rope_conjugate; B.qk_norm = c.qk_norm; B.single_stream = c.max_streams == 1 ? 1 : 0;
    B.gqa = (unsigned)(e->Hs / e->KVs); B.gqa_inv = udiv_inv(B.gqa);
    B.x = x_in; B.normw = L.attn_norm; B.eps = c.rms_eps; B.scale = (float)(1.0 / std::sqrt((double)e->hd));
    B.rope_cos = e->rope_cos; B.rope_sin = e->rope_sin;
    B.kcache = e->kcache + (long long)l * e->kv_layer_stride; B.vcache = e->vcache + (long long)l * e->kv_layer_stride;
    B.kv_stream_stride = e->kv_stream_stride; B.ctl = e->ctl;
    B.bias_q = L.bq; B.bias_k = L.bk; B.bias_v = L.bv; B.bias_out = L.bo; B.parts = e->parts;
    B.xchg = e->xchg; B.tick = e->tick; B.layer_tag = (unsigned)(l + 1); B.status = e->tick + 1; B.host_status = e->h_status; B.spin_limit = e->spin_limit;
    return B;
}

// Small models at short contexts, both halves of a layer fused (nl_block.h): embed -> { attn_block -> ffn_block } x L ->
// lm_head -> argmax, 2 launches per layer.  Each block adds its predecessor's partial vectors to the residual stream in
// its prologue (one designated workgroup stores the sum): the feed-forward block reads x[0] and writes x[1], the
// attention block of the next layer reads x[1] and writes x[0]; the embedding row lands in x[0].
void build_plan_blocks(nl_engine *e, std::vector<Op> &plan) {
    plan.clear();
    const nl_config &c = e->cfg;
    const int nslices = e->Is / FFN_SLICE;
    {
        EmbedParams P{e->embd_raw, e->embd_type, c.dim, e->ctl, e->x[0], e->gamma_row, e->gamma_val, e->tick};
        e->plan_embed = P;
        plan.push_back({K_EMBED, 0, nullptr, 0, [P](hipStream_t st) {
                               hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, st, P);
                               return hipGetLastError();
                           }});
    }
    for (int l = 0; l < c.n_layers; l++) {
        nl_engine::Layer &L = e->layers[l];
        {
            BlockParams B = attn_block_params(e, l, l == 0 ? e->x[0] : e->x[1]);
            if (l > 0) { B.parts_in = e->parts_ffn; B.nparts_in = nslices; B.x_out = e->x[0]; }
            const int wt = L.qkv.wtype, grid = blk_grid(e->Hs), npin = l > 0 ? nslices : 0;
            const size_t lds = blk_lds_bytes(c.dim);
            plan.push_back({K_ATTNBLOCK, 0, nullptr, 0, [B, wt, grid, lds, npin, e](hipStream_t st) mutable {
                                   B.dbg = e->dbg_block;      // nl_debug_stamps only
                                   return wt == WT_Q8_0 ? launch_attn_block<WT_Q8_0>(B, npin, grid, lds, st)
                                                        : launch_attn_block<WT_Q4_0>(B, npin, grid, lds, st);
                               }});
        }
        if (l == c.n_layers - 1) {
            // last layer: gate/up and down as the two GEMV launches, so that the residual stream is complete in memory.
            // (Measured: the feed-forward block here too, with the LM head adding the I / 256 partial vectors in its
            // prologue -- 28 launches instead of 30 -- costs the LM head what it saves: its ~1000 workgroups each re-read
            // seven vectors from L2, 6.5 -> 8.6 us, nano 5440 -> 5424 tok/s.)
            GemvParams P = base_params(e, L.gate, 2);
            P.q1 = L.up.q; P.s1 = L.up.s;
            P.x = e->x[0]; P.normw = L.ffn_norm; P.out = e->hb;
            P.parts = e->parts; P.nparts = e->Hs; P.x_out = e->x[1];
            push_gemv(plan, K_GATEUP, 0, nullptr, 0, L.gate.wtype, PRO_NORM_PARTS, EPI_SWIGLU, P);
            GemvParams Q = base_params(e, L.down);
            Q.x = e->hb; Q.out = e->x[1]; Q.resid = e->x[1];
            push_gemv(plan, K_DOWN, 0, nullptr, 0, L.down.wtype, PRO_PLAIN, EPI_RESID, Q);
        } else {
            FfnParams F{};
            F.gate_q = L.gate.q; F.gate_s = L.gate.s; F.up_q = L.up.q; F.up_s = L.up.s; F.dn_q = L.dn_slice.q; F.dn_s = L.dn_slice.s;
            F.D = c.dim; F.I = e->Is; F.npairs = L.gate.npairs;
            F.x = e->x[0]; F.normw = L.ffn_norm; F.parts_in = e->parts; F.nparts_in = e->Hs; F.x_out = e->x[1]; F.eps = c.rms_eps;
            F.parts_out = e->parts_ffn; F.xchg = e->xchg_ffn; F.tick = e->tick; F.layer_tag = (unsigned)(l + 1);
            F.status = e->tick + 1; F.host_status = e->h_status; F.spin_limit = e->spin_limit;
            const int wt = L.gate.wtype, grid = ffn_grid(nslices);
            plan.push_back({K_FFNBLOCK, 0, nullptr, 0, [F, wt, grid, e](hipStream_t st) mutable {
                                   F.dbg = e->dbg_block;      // nl_debug_stamps only
                                   return wt == WT_Q8_0 ? launch_ffn_block<WT_Q8_0>(F, grid, st) : launch_ffn_block<WT_Q4_0>(F, grid, st);
                               }});
        }
    }
    int lm_blocks, lm_spb;
    {   // final RMSNorm + LM head (go/model.go:616-619)
        GemvParams P = base_params(e, e->lm_head);
        P.x = e->x[1]; P.normw = e->output_norm;
        P.out = e->logits; P.amax_val = e->amax_val; P.amax_idx = e->amax_idx;
        P.host_out = e->d_h_logits;
        lm_blocks = (P.ntiles + P.tw - 1) / P.tw;
        lm_spb = (P.tw * TR + 63) / 64;
        push_gemv(plan, K_LMHEAD, 0, e->logits, (size_t)e->Vs, e->lm_head.wtype, PRO_NORM, EPI_STORE, P);
    }
    {
        ArgmaxParams P{e->logits, c.vocab, e->amax_val, e->amax_idx, lm_blocks * lm_spb, e->ctl, e->ids, e->result};
        e->plan_argmax = P;
        plan.push_back({K_ARGMAX, 0, nullptr, 0, [P](hipStream_t st) {
                               hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, st, P);
                               return hipGetLastError();
                           }});
    }
    link_prefetch(plan);
}

// Build the per-token launch plan: the device-side restatement of Forward
// (go/model.go:490-620).  token / pos / stream are read from e->ctl by the
// kernels, so one captured graph serves every step.
void build_plan(nl_engine *e, std::vector<Op> &plan, bool fused, bool help = false) {
    if (fused && e->fused_mode == 1 && e->ffn_fused) { build_plan_blocks(e, plan); return; }
    plan.clear();
    const nl_config &c = e->cfg;
    const bool p2p = e->p2p.on;
    const bool tp = (e->G > 1 || e->force_tp_plan) && !p2p;   // RCCL seams (or the in-process group's own sums)
    int cur = 0;
    int lm_blocks = 0, lm_spb = 1;
    const float *pending = nullptr;  // all-reduced partial still to be added to the residual stream
    // push seams (nl_p2p.h): the partial leaves the producing GEMV as granules into every rank's receive slots and the
    // same launch finishes the all-reduce (each row's owner lane adds the G granules of its row to the residual stream,
    // EPI_P2P).  Slots alternate by seam parity.  Loopback (nl_p2p_loopback: one rank alone, timing only): the rank
    // writes its partial into ALL G rank-slots of its own area, so grids, stores and polls are the real ones.
    const size_t slot_bytes = (size_t)c.dim * sizeof(u64);
    auto p2p_producer = [&](GemvParams &P, int seam, float *x) {
        for (int r = 0; r < e->G; r++)
            P.p2p_dst[r] = reinterpret_cast<u64 *>((char *)e->p2p.peer[r] + ((size_t)(seam & 1) * e->G + (e->p2p.loopback ? r : e->rank)) * slot_bytes);
        P.p2p_slots = reinterpret_cast<const u64 *>((char *)e->p2p.area + (size_t)(seam & 1) * e->G * slot_bytes);
        P.p2p_n = e->G; P.p2p_epoch = e->p2p.epoch; P.p2p_seam = (unsigned)(seam + 1);
        P.p2p_status = e->p2p.status; P.p2p_timeout = e->p2p.timeout_ticks;
        P.out = x; P.resid = x;
    };
    int seam = 0;

    {
        EmbedParams P{e->embd_raw, e->embd_type, c.dim, e->ctl, e->x[0], e->gamma_row, e->gamma_val, p2p ? e->p2p.epoch : e->tick};
        e->plan_embed = P;
        plan.push_back({K_EMBED, 0, nullptr, 0, [P](hipStream_t st) {
                               hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, st, P);
                               return hipGetLastError();
                           }});
    }
    for (int l = 0; l < c.n_layers; l++) {
        nl_engine::Layer &L = e->layers[l];
        float *kc = e->kcache + (long long)l * e->kv_layer_stride;
        float *vc = e->vcache + (long long)l * e->kv_layer_stride;
        bool parts_pending = false;
        // nl_tp.h's attention half (projection + RoPE + KV store + attention + WO) as one op: a tensor-parallel rank's (mode 3:
        // the seam's all-reduce in its tail) or, with a direct seam, the whole layer's on one GPU (mode 4)
        auto tp_attn_op = [&](const TpSeam &sm, int coll, float *cbuf) {
        TpAttnParams Q{};
        GroupParams &B = Q.G;
        B.qkv_q = L.qkv.q; B.qkv_s = L.qkv.s;
        B.D = c.dim; B.npairs = L.qkv.npairs; B.n_q_heads = e->Hs; B.n_kv_heads = e->KVs; B.seq_len = c.seq_len;
        B.rope_conj = c.rope_conjugate; B.qk_norm = c.qk_norm; B.single_stream = c.max_streams == 1 ? 1 : 0;
        B.tpm = e->grp_tpm; B.members = (e->gqa + 2) * 4 / e->grp_tpm;
        B.gqa = (unsigned)e->gqa; B.wpt = (unsigned)((GRP_THREADS / 64) / e->grp_tpm); B.wpt_inv = udiv_inv(B.wpt);
        B.m8_inv = udiv_inv(8u * (unsigned)B.members); B.m_inv = udiv_inv((unsigned)B.members);
        B.x = e->x[cur]; B.normw = L.attn_norm; B.eps = c.rms_eps; B.scale = (float)(1.0 / std::sqrt((double)e->hd));
        B.rope_cos = e->rope_cos; B.rope_sin = e->rope_sin; B.kcache = kc; B.vcache = vc;
        B.kv_stream_stride = e->kv_stream_stride; B.ctl = e->ctl;
        B.bias_q = L.bq; B.bias_k = L.bk; B.bias_v = L.bv;
        B.nsplit_max = e->nsplit_max;
        B.xchg = e->xchg; B.tick = p2p ? e->p2p.epoch : e->tick; B.layer_tag = (unsigned)(l + 1);
        B.status = e->tick + 1; B.host_status = e->h_status; B.spin_limit = e->spin_limit;
        Q.wo_q = L.wo.q; Q.wo_s = L.wo.s; Q.wo_npairs = L.wo.npairs; Q.wo_ntiles = L.wo.ntiles; Q.wo_gshift = e->tpg.wo_gshift;
        Q.wo_tpw = e->tpg.wo_tpw;
        Q.n_heads_local = e->Hs; Q.xq = e->tp_xq; Q.xo = e->tp_xo; Q.bias_out = L.bo; Q.x = e->x[cur];
        Q.xp = e->tp_xp; Q.helpers = (help && e->attn_helpers) ? 1 : 0; Q.live_grid = grp_grid(e->KVs, B.members);
        Q.seam = sm;
        const bool mfa = e->mf_attn;
        if (mfa) { B.qkv_q = L.qkv.q2; B.qkv_s = L.qkv.s2; }       // (WO stays on the vector pipe: nl_tp.h)
        if (e->wide_ffn && (e->prefetch & 1) && L.gate.wtype == L.up.wtype && (L.gate.wtype == WT_Q4_0 || L.gate.wtype == WT_Q8_0)) {
            Q.pf.q[0] = e->mf_ffn ? L.gate.q2 : L.gate.q; Q.pf.q[1] = e->mf_ffn ? L.up.q2 : L.up.q;
            Q.pf.s[0] = e->mf_ffn ? L.gate.s2 : L.gate.s; Q.pf.s[1] = e->mf_ffn ? L.up.s2 : L.up.s;
            Q.pf.tile_qbytes = tile_qbytes(L.gate.wtype, L.gate.npairs); Q.pf.tile_sbytes = tile_sbytes(L.gate.wtype, L.gate.npairs);
            Q.pf.ntiles = L.gate.ntiles; Q.pf.nmat = 2; Q.pf.blocks = L.down.ntiles; Q.pf_early = (e->prefetch & 8) ? 1 : 0;
        }
        if ((e->prefetch & 16) && Q.pf.nmat > 0 && l + 1 < c.n_layers) {
            const nl_engine::Layer &N = e->layers[l + 1];
            if (N.qkv.wtype == WT_Q4_0 || N.qkv.wtype == WT_Q8_0) {
                Q.pf2.q[0] = Q.pf2.q[1] = e->mf_attn ? N.qkv.q2 : N.qkv.q; Q.pf2.s[0] = Q.pf2.s[1] = e->mf_attn ? N.qkv.s2 : N.qkv.s;
                Q.pf2.tile_qbytes = tile_qbytes(N.qkv.wtype, N.qkv.npairs); Q.pf2.tile_sbytes = tile_sbytes(N.qkv.wtype, N.qkv.npairs);
                Q.pf2.ntiles = N.qkv.ntiles; Q.pf2.nmat = 1; Q.pf2.blocks = 0;
            }
        }
        const int wt = L.qkv.wtype, grid = std::max(grp_grid(e->KVs, B.members), (L.wo.ntiles + e->tpg.wo_tpw - 1) / e->tpg.wo_tpw);
        const int ngroups = (L.qkv.npairs + KL - 1) / KL, nf = (ngroups + 16 / e->grp_tpm - 1) / (16 / e->grp_tpm);
        const size_t lds = tp_attn_lds_bytes(L.wo.npairs);
        const bool hlp = Q.helpers != 0;
        Op op{K_ATTNBLOCK, coll, cbuf, (size_t)c.dim, [Q, wt, grid, nf, lds, mfa, hlp](hipStream_t st) {
                  if (hlp && nf == 1) hipLaunchKernelGGL((tp_attn_kernel<WT_Q4_0, 1, false, true>), dim3(grid), dim3(TP_THREADS), lds, st, Q);
                  else if (hlp) hipLaunchKernelGGL((tp_attn_kernel<WT_Q4_0, 2, false, true>), dim3(grid), dim3(TP_THREADS), lds, st, Q);
                  else if (mfa && nf == 1) hipLaunchKernelGGL((tp_attn_kernel<WT_Q4_0, 1, true>), dim3(grid), dim3(TP_THREADS), lds, st, Q);
                  else if (mfa) hipLaunchKernelGGL((tp_attn_kernel<WT_Q4_0, 2, true>), dim3(grid), dim3(TP_THREADS), lds, st, Q);
                  else if (wt == WT_Q8_0 && nf == 1) hipLaunchKernelGGL((tp_attn_kernel<WT_Q8_0, 1>), dim3(grid), dim3(TP_THREADS), lds, st, Q);
                  else if (wt == WT_Q8_0) hipLaunchKernelGGL((tp_attn_kernel<WT_Q8_0, 2>), dim3(grid), dim3(TP_THREADS), lds, st, Q);
                  else if (nf == 1) hipLaunchKernelGGL((tp_attn_kernel<WT_Q4_0, 1>), dim3(grid), dim3(TP_THREADS), lds, st, Q);
                  else hipLaunchKernelGGL((tp_attn_kernel<WT_Q4_0, 2>), dim3(grid), dim3(TP_THREADS), lds, st, Q);
                  return hipGetLastError();
              }};
        op.add_to = e->x[cur];
        return op;
        };
        // the feed-forward half as one launch of wide_ffn_kernel (nl_tp.h): one GPU (direct seam) or a tensor-parallel rank's
        auto wide_ffn_op = [&](const TpSeam &sm, int coll, float *cbuf) {
        WideFfnParams F{};
        F.gate_q = L.gate.q; F.gate_s = L.gate.s; F.up_q = L.up.q; F.up_s = L.up.s; F.dn_q = L.down.q; F.dn_s = L.down.s;
        F.D = c.dim; F.I = e->Is; F.npairs = L.gate.npairs; F.gu_tiles = L.gate.ntiles;
        F.dn_npairs = L.down.npairs; F.dn_ntiles = L.down.ntiles;
        const int grid = L.down.ntiles;
        F.rounds = (L.gate.ntiles + grid - 1) / grid;
        F.normw = L.ffn_norm; F.eps = c.rms_eps; F.x = e->x[cur]; F.hx = e->tp_hx;
        F.tick = p2p ? e->p2p.epoch : e->tick; F.layer_tag = (unsigned)(l + 1);
        F.status = e->tick + 1; F.host_status = e->h_status; F.spin_limit = e->spin_limit;
        const int wt = L.gate.wtype, nf = e->wide_nf, ngc = e->wide_ngc, rounds = F.rounds;
        const size_t lds = wide_ffn_lds_bytes(nf, L.down.npairs);
        F.seam = sm;
        const bool mff = e->mf_ffn;
        if (mff) { F.gate_q = L.gate.q2; F.gate_s = L.gate.s2; F.up_q = L.up.q2; F.up_s = L.up.s2; F.dn_q = L.down.q2; F.dn_s = L.down.s2; }
        F.pf_ahead = (e->prefetch & 4) ? 1 : 0;
        if ((e->prefetch & 2) && fused && (e->fused_mode == 3 || e->fused_mode == 4) && l + 1 < c.n_layers) {
            const nl_engine::Layer &N = e->layers[l + 1];
            if (N.qkv.wtype == WT_Q4_0 || N.qkv.wtype == WT_Q8_0) {
                PfQkv &X = F.pf;
                X.T.q[0] = X.T.q[1] = e->mf_attn ? N.qkv.q2 : N.qkv.q; X.T.s[0] = X.T.s[1] = e->mf_attn ? N.qkv.s2 : N.qkv.s;
                X.T.tile_qbytes = tile_qbytes(N.qkv.wtype, N.qkv.npairs); X.T.tile_sbytes = tile_sbytes(N.qkv.wtype, N.qkv.npairs);
                X.T.ntiles = N.qkv.ntiles; X.T.nmat = 1; X.T.blocks = grid;
                X.tpm = e->grp_tpm; X.members = (e->gqa + 2) * 4 / e->grp_tpm; X.gqa = e->gqa; X.n_q_heads = e->Hs; X.n_kv_heads = e->KVs;
                X.m8_inv = udiv_inv(8u * (unsigned)X.members); X.m_inv = udiv_inv((unsigned)X.members);
            }
        }
        Op op{K_FFNBLOCK, coll, cbuf, (size_t)c.dim, [F, wt, grid, nf, ngc, rounds, lds, mff](hipStream_t st) {
#define NL_WF(WT_, NF_, NGC_, R_) do { if (mff && WT_ == WT_Q4_0) hipLaunchKernelGGL((wide_ffn_kernel<WT_Q4_0, NF_, NGC_, R_, true>), dim3(grid), dim3(TP_THREADS), lds, st, F); \
                                       else hipLaunchKernelGGL((wide_ffn_kernel<WT_, NF_, NGC_, R_>), dim3(grid), dim3(TP_THREADS), lds, st, F); } while (0)
#define NL_WF1(WT_, NF_, NGC_) do { if (rounds <= 1) NL_WF(WT_, NF_, NGC_, 1); else if (rounds == 2) NL_WF(WT_, NF_, NGC_, 2); \
                            else if (rounds == 3) NL_WF(WT_, NF_, NGC_, 3); else NL_WF(WT_, NF_, NGC_, 4); } while (0)
#define NL_WF2(WT_, NF_) do { if (ngc <= 1) NL_WF1(WT_, NF_, 1); else NL_WF1(WT_, NF_, 3); } while (0)
                               if (wt == WT_Q8_0) { if (nf == 1) NL_WF2(WT_Q8_0, 1); else NL_WF2(WT_Q8_0, 2); }
                               else { if (nf == 1) NL_WF2(WT_Q4_0, 1); else NL_WF2(WT_Q4_0, 2); }
#undef NL_WF2
#undef NL_WF1
#undef NL_WF
                               return hipGetLastError();
                           }};
        op.add_to = e->x[cur];
        return op;
        };
        if (fused && e->fused_mode == 3) {
            // a tensor-parallel rank's layer as two launches (nl_tp.h): both finish their all-reduce seam in the tail (push
            // path), or leave the rank's partial in `ar` for the in-process group to sum and add (coll 3)
            auto make_seam = [&](int sm) {
                TpSeam S{};
                S.rows = c.dim;
                if (p2p) {
                    for (int r = 0; r < e->G; r++)
                        S.dst[r] = reinterpret_cast<u64 *>((char *)e->p2p.peer[r] + ((size_t)(sm & 1) * e->G + (e->p2p.loopback ? r : e->rank)) * slot_bytes);
                    S.slots = reinterpret_cast<const u64 *>((char *)e->p2p.area + (size_t)(sm & 1) * e->G * slot_bytes);
                    S.n = e->G; S.epoch = e->p2p.epoch; S.seam = (unsigned)(sm + 1);
                    S.status = e->p2p.status; S.timeout = e->p2p.timeout_ticks;
                } else {
                    S.n = 0; S.partial = e->ar;
                }
                return S;
            };
            {
                plan.push_back(tp_attn_op(make_seam(seam++), p2p ? 0 : 3, p2p ? nullptr : e->ar));
            }
            if (e->wide_ffn) {
                plan.push_back(wide_ffn_op(make_seam(seam++), p2p ? 0 : 3, p2p ? nullptr : e->ar));
                continue;
            }
            {
                TpFfnParams F{};
                F.gate_q = L.gate.q; F.gate_s = L.gate.s; F.up_q = L.up.q; F.up_s = L.up.s; F.dn_q = L.down.q; F.dn_s = L.down.s;
                F.D = c.dim; F.I = e->Is; F.npairs = L.gate.npairs; F.gu_tiles = L.gate.ntiles;
                F.dn_npairs = L.down.npairs; F.dn_ntiles = L.down.ntiles;
                F.pair = e->tpg.pair; F.n_prod = e->tpg.n_prod; F.n_cons = e->tpg.n_cons; F.ct_shift = e->tpg.ct_shift;
                F.normw = L.ffn_norm; F.eps = c.rms_eps; F.x = e->x[cur]; F.hx = e->tp_hx;
                F.tick = p2p ? e->p2p.epoch : e->tick; F.layer_tag = (unsigned)(l + 1);
                F.status = e->tick + 1; F.host_status = e->h_status; F.spin_limit = e->spin_limit;
                F.seam = make_seam(seam++);
                const int wt = L.gate.wtype, grid = e->tpg.grid, nf = e->tpg.nf, ngc = e->tpg.ngc, nr = e->tpg.nr;
                const size_t lds = tp_ffn_lds_bytes(L.down.npairs);
                Op op{K_FFNBLOCK, p2p ? 0 : 3, p2p ? nullptr : e->ar, (size_t)c.dim, [F, wt, grid, nf, ngc, nr, lds](hipStream_t st) {
#define NL_TPF(WT_, NF_, NGC_, NR_) hipLaunchKernelGGL((tp_ffn_kernel<WT_, NF_, NGC_, NR_>), dim3(grid), dim3(TP_THREADS), lds, st, F)
#define NL_TPF2(WT_, NF_, NGC_) do { if (nr <= 1) NL_TPF(WT_, NF_, NGC_, 1); else NL_TPF(WT_, NF_, NGC_, 2); } while (0)
#define NL_TPF3(WT_, NF_) do { if (ngc <= 1) NL_TPF2(WT_, NF_, 1); else NL_TPF2(WT_, NF_, 2); } while (0)
                          if (wt == WT_Q8_0) { if (nf == 1) NL_TPF3(WT_Q8_0, 1); else NL_TPF3(WT_Q8_0, 2); }
                          else { if (nf == 1) NL_TPF3(WT_Q4_0, 1); else NL_TPF3(WT_Q4_0, 2); }
#undef NL_TPF3
#undef NL_TPF2
#undef NL_TPF
                          return hipGetLastError();
                      }};
                op.add_to = e->x[cur];
                plan.push_back(op);
            }
            continue;
        }
        if (fused && e->fused_mode == 1) {
            // the whole attention half as one launch per layer (nl_block.h); its H partial vectors are added to the
            // residual stream by the gate/up prologue below
            BlockParams B = attn_block_params(e, l, e->x[cur]);
            const int wt = L.qkv.wtype, grid = blk_grid(e->Hs);
            const size_t lds = blk_lds_bytes(c.dim);
            plan.push_back({K_ATTNBLOCK, 0, nullptr, 0, [B, wt, grid, lds, e](hipStream_t st) mutable {
                                   B.dbg = e->dbg_block;      // nl_debug_stamps only
                                   return wt == WT_Q8_0 ? launch_attn_block<WT_Q8_0>(B, 0, grid, lds, st)
                                                        : launch_attn_block<WT_Q4_0>(B, 0, grid, lds, st);
                               }});
            parts_pending = true;
        } else if (fused && e->fused_mode == 4) {
            // one GPU, wide tier: projection + RoPE + KV store + attention + WO + residual as ONE launch (nl_tp.h with a direct
            // seam); gate / up and down follow as GEMVs on the finished residual stream
            TpSeam S{};
            S.n = -1; S.rows = c.dim;
            plan.push_back(tp_attn_op(S, 0, nullptr));
            if (e->wide_ffn) {
                plan.push_back(wide_ffn_op(S, 0, nullptr));
                continue;
            }
        } else {
        if (fused && e->fused_mode == 2) {
            // projection + RoPE + KV store + attention as one launch (nl_group.h); WO below consumes its partials
            GroupParams B{};
            B.qkv_q = L.qkv.q; B.qkv_s = L.qkv.s;
            B.D = c.dim; B.npairs = L.qkv.npairs; B.n_q_heads = e->Hs; B.n_kv_heads = e->KVs; B.seq_len = c.seq_len;
            B.rope_conj = c.rope_conjugate; B.qk_norm = c.qk_norm; B.single_stream = c.max_streams == 1 ? 1 : 0;
            B.tpm = e->grp_tpm; B.members = (e->gqa + 2) * 4 / e->grp_tpm;
            B.gqa = (unsigned)e->gqa; B.wpt = (unsigned)((GRP_THREADS / 64) / e->grp_tpm); B.wpt_inv = udiv_inv(B.wpt);
            B.m8_inv = udiv_inv(8u * (unsigned)B.members); B.m_inv = udiv_inv((unsigned)B.members);
            B.x = e->x[cur]; B.normw = L.attn_norm; B.eps = c.rms_eps; B.scale = (float)(1.0 / std::sqrt((double)e->hd));
            B.rope_cos = e->rope_cos; B.rope_sin = e->rope_sin; B.kcache = kc; B.vcache = vc;
            B.kv_stream_stride = e->kv_stream_stride; B.ctl = e->ctl;
            B.bias_q = L.bq; B.bias_k = L.bk; B.bias_v = L.bv;
            B.part_o = e->part_o; B.part_ml = e->part_ml; B.nsplit_max = e->nsplit_max;
            B.xchg = e->xchg; B.tick = p2p ? e->p2p.epoch : e->tick; B.layer_tag = (unsigned)(l + 1);
            B.status = e->tick + 1; B.host_status = e->h_status; B.spin_limit = e->spin_limit;
            const int wt = L.qkv.wtype, grid = grp_grid(e->KVs, B.members);
            const int ngroups = (L.qkv.npairs + KL - 1) / KL, nf = (ngroups + 16 / e->grp_tpm - 1) / (16 / e->grp_tpm);
            plan.push_back({K_ATTNBLOCK, 0, nullptr, 0, [B, wt, grid, nf](hipStream_t st) {
                                   const size_t lds = grp_lds_bytes();
                                   if (wt == WT_Q8_0 && nf == 1) hipLaunchKernelGGL((qkv_attn_kernel<WT_Q8_0, 1>), dim3(grid), dim3(GRP_THREADS), lds, st, B);
                                   else if (wt == WT_Q8_0) hipLaunchKernelGGL((qkv_attn_kernel<WT_Q8_0, 2>), dim3(grid), dim3(GRP_THREADS), lds, st, B);
                                   else if (nf == 1) hipLaunchKernelGGL((qkv_attn_kernel<WT_Q4_0, 1>), dim3(grid), dim3(GRP_THREADS), lds, st, B);
                                   else hipLaunchKernelGGL((qkv_attn_kernel<WT_Q4_0, 2>), dim3(grid), dim3(GRP_THREADS), lds, st, B);
                                   return hipGetLastError();
                               }});
        } else {
        {   // RMSNorm + Q,K,V GEMV + RoPE + KV store   (go/model.go:517-554)
            GemvParams P = base_params(e, L.qkv);
            P.x = e->x[cur]; P.normw = L.attn_norm;
            if (pending) { P.add = pending; P.x_out = e->x[cur ^ 1]; }
            P.rope_cos = e->rope_cos; P.rope_sin = e->rope_sin;
            P.qbuf = e->qbuf; P.kcache = kc; P.vcache = vc; P.kv_stream_stride = e->kv_stream_stride;
            P.n_q_heads = e->Hs; P.n_kv_heads = e->KVs; P.seq_len = c.seq_len; P.rope_conj = c.rope_conjugate;
            P.bias_q = L.bq; P.bias_k = L.bk; P.bias_v = L.bv;
            int wt = L.qkv.wtype;
            push_gemv(plan, K_QKV, 0, nullptr, 0, wt, PRO_NORM, EPI_QKV, P);
            if (pending) { cur ^= 1; pending = nullptr; }
        }
        if (c.qk_norm) {
            QkNormParams P{e->qbuf, kc, e->kv_stream_stride, e->ctl, e->Hs, e->KVs, e->hd, c.seq_len, c.rms_eps};
            int nh = e->Hs + e->KVs;
            plan.push_back({K_QKV, 0, nullptr, 0, [P, nh](hipStream_t st) {
                                   hipLaunchKernelGGL(qknorm_kernel, dim3(nh), dim3(64), 0, st, P);
                                   return hipGetLastError();
                               }});
        }
        {   // GQA attention over the cache (go/model.go:557-587)
            AttnParams P{e->qbuf, kc, vc, e->kv_stream_stride, e->part_o, e->part_ml, e->ctl,
                         e->KVs, c.seq_len, e->nsplit_max, (float)(1.0 / std::sqrt((double)e->hd)), c.max_streams == 1 ? 1 : 0,
                         nullptr, nullptr, 0, 0};
            dim3 grid(e->KVs, e->nsplit_max);
            int hd = e->hd, gqa = e->gqa;
            plan.push_back({K_ATTN, 0, nullptr, 0,
                               [hd, gqa, P, grid](hipStream_t st) { return launch_attn(hd, gqa, P, grid, st); }});
        }
        }   // five-launch projection / attention
        {   // WO + residual (go/model.go:590-594); the prologue merges the attention splits
            GemvParams P = base_params(e, L.wo);
            P.part_o = e->part_o; P.part_ml = e->part_ml;
            P.bias_out = L.bo;
            int wt = L.wo.wtype;
            if (p2p) {
                p2p_producer(P, seam++, e->x[cur]);
                push_gemv(plan, K_WO, 0, nullptr, 0, wt, PRO_ATTN, EPI_P2P, P);
            } else if (!tp) {
                P.out = e->x[cur]; P.resid = e->x[cur];
                push_gemv(plan, K_WO, 0, nullptr, 0, wt, PRO_ATTN, EPI_RESID, P);
            } else {
                P.out = e->ar;
                push_gemv(plan, K_WO, 1, e->ar, (size_t)c.dim, wt, PRO_ATTN, EPI_STORE, P);
                pending = e->ar;
            }
        }
        }   // !fused
        {   // RMSNorm + gate/up GEMV + SiLU*up (go/model.go:597-606)
            GemvParams P = base_params(e, L.gate, 2);
            P.q1 = L.up.q; P.s1 = L.up.s;
            P.x = e->x[cur]; P.normw = L.ffn_norm; P.out = e->hb;
            if (pending) { P.add = pending; P.x_out = e->x[cur ^ 1]; }
            int wt = L.gate.wtype;
            if (parts_pending) {
                P.parts = e->parts; P.nparts = e->Hs; P.x_out = e->x[cur ^ 1];
                push_gemv(plan, K_GATEUP, 0, nullptr, 0, wt, PRO_NORM_PARTS, EPI_SWIGLU, P);
                cur ^= 1;
            } else {
                push_gemv(plan, K_GATEUP, 0, nullptr, 0, wt, PRO_NORM, EPI_SWIGLU, P);
            }
            if (pending) { cur ^= 1; pending = nullptr; }
        }
        {   // down + residual (go/model.go:609-612)
            GemvParams P = base_params(e, L.down);
            P.x = e->hb;
            int wt = L.down.wtype;
            if (p2p) {
                p2p_producer(P, seam++, e->x[cur]);
                push_gemv(plan, K_DOWN, 0, nullptr, 0, wt, PRO_PLAIN, EPI_P2P, P);
            } else if (!tp) {
                P.out = e->x[cur]; P.resid = e->x[cur];
                push_gemv(plan, K_DOWN, 0, nullptr, 0, wt, PRO_PLAIN, EPI_RESID, P);
            } else {
                P.out = e->ar;
                push_gemv(plan, K_DOWN, 1, e->ar, (size_t)c.dim, wt, PRO_PLAIN, EPI_STORE, P);
                pending = e->ar;
            }
        }
    }
    {   // final RMSNorm + LM head (go/model.go:616-619)
        GemvParams P = base_params(e, e->lm_head);
        P.x = e->x[cur]; P.normw = e->output_norm;
        if (pending) { P.add = pending; P.x_out = e->x[cur ^ 1]; }
        P.out = e->logits + (size_t)e->rank * e->Vs;
        if (!tp) { P.amax_val = e->amax_val; P.amax_idx = e->amax_idx; P.host_out = e->d_h_logits; }
        if (p2p)   // the logits all-gather: this rank's slice also lands in every peer's gathered buffer
            for (int r = 0; r < e->G; r++)
                if (r != e->rank)   // (loopback: the slice fills every other rank's place in this rank's own buffer)
                    P.peer_out[r] = reinterpret_cast<float *>((char *)e->p2p.peer[r] + e->p2p.off_logits) + (size_t)(e->p2p.loopback ? r : e->rank) * e->Vs;
        int wt = e->lm_head.wtype;
        lm_blocks = (P.ntiles + P.tw - 1) / P.tw;
        lm_spb = (P.tw * TR + 63) / 64;
        push_gemv(plan, K_LMHEAD, tp ? 2 : 0, e->logits, (size_t)e->Vs, wt, PRO_NORM, EPI_STORE, P);
    }
    if (p2p) {
        P2PArgmaxParams P{};
        P.A = ArgmaxParams{e->logits, c.vocab, e->amax_val, e->amax_idx, lm_blocks * lm_spb, e->ctl, e->ids, e->result};
        for (int r = 0; r < e->G; r++)
            P.dst[r] = reinterpret_cast<u64 *>((char *)e->p2p.peer[r] + e->p2p.off_amax) + 4 * (e->p2p.loopback ? r : e->rank);
        P.slots = reinterpret_cast<const u64 *>((char *)e->p2p.area + e->p2p.off_amax);
        // a fused-launch give-up on ANY rank travels with the pair: every rank retires its fused plan in the same call
        if (e->tick && e->h_status) { P.fstatus = e->tick + 1; P.host_fstatus = e->h_status; }
        P.G = e->G; P.row0 = e->rank * e->Vs; P.seam = 255u;
        P.epoch = e->p2p.epoch; P.status = e->p2p.status; P.timeout_ticks = e->p2p.timeout_ticks;
        e->plan_p2p_argmax = P;
        plan.push_back({K_ARGMAX, 0, nullptr, 0, [P](hipStream_t st) {
                               hipLaunchKernelGGL(p2p_argmax_kernel, dim3(1), dim3(1024), 0, st, P);
                               return hipGetLastError();
                           }});
    } else {
        ArgmaxParams P{e->logits, c.vocab, tp ? nullptr : e->amax_val, e->amax_idx, lm_blocks * lm_spb, e->ctl, e->ids,
                       e->result};
        e->plan_argmax = P;
        plan.push_back({K_ARGMAX, 0, nullptr, 0, [P](hipStream_t st) {
                               hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, st, P);
                               return hipGetLastError();
                           }});
    }
    link_prefetch(plan);
}

int run_collective(nl_engine *e, const Op &op) {
    if (!op.coll) return NL_OK;
    if (!e->comm) return e->fail(NL_ERR_COMM, "tensor-parallel forward without nl_comm_init (local groups use nl_group_forward)");
    int rc;
    if (op.coll == 1) rc = g_rccl.AllReduce(op.buf, op.buf, op.count, kNcclFloat, kNcclSum, e->comm, e->stream);
    else rc = g_rccl.AllGather(op.buf + (size_t)e->rank * op.count, op.buf, op.count, kNcclFloat, e->comm, e->stream);
    if (rc != 0) return e->fail(NL_ERR_COMM, "rccl collective failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return NL_OK;
}

int run_plan_eager(nl_engine *e, const nl_engine::PlanSet &S) {
    for (const Op &op : S.ops) {
        hipError_t s = op.fn(e->stream);
        if (s != hipSuccess) return e->fail(NL_ERR_HIP, "launch %s: %s", kKindNames[op.kind], hipGetErrorString(s));
        int rc = run_collective(e, op);
        if (rc) return rc;
    }
    return NL_OK;
}

__global__ void box_fetch_kernel(const int *box, int *ctl) {
    if (threadIdx.x < CTL_WORDS) ctl[threadIdx.x] = __hip_atomic_load(box + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void box_done_kernel(const int *box, const int *result, unsigned long long *done) {
    // (behind the LM head and the argmax of the step: their stores -- the logits rows in the pinned buffer too -- are complete)
    const unsigned seq = (unsigned)__hip_atomic_load(box + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(done, ((unsigned long long)seq << 32) | (unsigned)*result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

void destroy_graphs(nl_engine::PlanSet &S) {
    if (S.call_exec) { (void)hipGraphExecDestroy(S.call_exec); S.call_exec = nullptr; }
    if (S.call) { (void)hipGraphDestroy(S.call); S.call = nullptr; }
    if (S.exec) { (void)hipGraphExecDestroy(S.exec); S.exec = nullptr; }
    if (S.multi_exec) { (void)hipGraphExecDestroy(S.multi_exec); S.multi_exec = nullptr; }
    if (S.multi) { (void)hipGraphDestroy(S.multi); S.multi = nullptr; }
    if (S.graph) { (void)hipGraphDestroy(S.graph); S.graph = nullptr; }
}

int capture_graph(nl_engine *e, nl_engine::PlanSet &S) {
    HIPCK(e, hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    int rc = run_plan_eager(e, S);
    hipGraph_t g = nullptr;
    hipError_t s = hipStreamEndCapture(e->stream, &g);
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    if (s != hipSuccess) return e->fail(NL_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(s));
    S.graph = g;
    HIPCK(e, hipGraphInstantiate(&S.exec, S.graph, nullptr, nullptr, 0));
    if (e->d_box && e->G == 1 && !e->force_tp_plan && !e->p2p.on && !getenv("NL_NO_CALL_BOX")) {
        HIPCK(e, hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(box_fetch_kernel, dim3(1), dim3(64), 0, e->stream, e->d_box, e->ctl);
        const int rc3 = run_plan_eager(e, S);
        hipLaunchKernelGGL(box_done_kernel, dim3(1), dim3(1), 0, e->stream, e->d_box, e->result, e->d_done);
        hipGraph_t gc = nullptr;
        const hipError_t s3 = hipStreamEndCapture(e->stream, &gc);
        if (!rc3 && s3 == hipSuccess && gc && hipGraphInstantiate(&S.call_exec, gc, nullptr, nullptr, 0) == hipSuccess) S.call = gc;
        else { if (gc) (void)hipGraphDestroy(gc); S.call_exec = nullptr; (void)hipGetLastError(); }     // (the per-call path keeps copy + launch + synchronise)
    }
    // chained greedy decode replays a graph that holds the plan 16 times: the gap between two graph launches
    // (~8 us on this stack) is then paid once per 16 tokens (nano 3834 -> 3945 tok/s; 4 steps: 3905; 32 / 64: as 16)
    static const int steps = getenv("NL_GRAPH_STEPS") ? atoi(getenv("NL_GRAPH_STEPS")) : 16;   // developer knob (tools/)
    if (steps > 1 && ((e->G == 1 && !e->force_tp_plan) || e->p2p.on)) {
        HIPCK(e, hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
        int rc2 = NL_OK;
        // step k's argmax also embeds step k+1's token (argmax_embed_kernel): the embed launch exists in step 0 only
        static const bool fuse_embed = getenv("NL_NO_ARGMAX_EMBED") == nullptr;
        for (int k = 0; k < steps && !rc2; k++)
            for (const Op &op : S.ops) {
                if (fuse_embed && op.kind == K_EMBED && k > 0) continue;
                hipError_t ls;
                if (fuse_embed && op.kind == K_ARGMAX && e->p2p.on) {
                    P2PArgmaxParams PA = e->plan_p2p_argmax;
                    PA.E = e->plan_embed;     // the winner's embedding row opens the next step (and advances the counter)
                    hipLaunchKernelGGL(p2p_argmax_kernel, dim3(1), dim3(1024), 0, e->stream, PA);
                    ls = hipGetLastError();
                } else if (fuse_embed && op.kind == K_ARGMAX) {
                    hipLaunchKernelGGL(argmax_embed_kernel, dim3(1), dim3(1024), 0, e->stream, e->plan_argmax, e->plan_embed);
                    ls = hipGetLastError();
                } else ls = op.fn(e->stream);
                if (ls != hipSuccess) { rc2 = e->fail(NL_ERR_HIP, "multi-step capture: %s", hipGetErrorString(ls)); break; }
            }
        hipGraph_t gm = nullptr;
        hipError_t s2 = hipStreamEndCapture(e->stream, &gm);
        if (rc2 || s2 != hipSuccess) { if (gm) hipGraphDestroy(gm); return rc2 ? rc2 : e->fail(NL_ERR_HIP, "multi-step capture"); }
        S.multi = gm;
        HIPCK(e, hipGraphInstantiate(&S.multi_exec, gm, nullptr, nullptr, 0));
        e->graph_steps = steps;
    }
    return NL_OK;
}

void destroy_samp_graphs(nl_engine *e) {
    for (int k = 0; k < 3; k++) {
        if (e->samp_graph_exec[k]) { (void)hipGraphExecDestroy(e->samp_graph_exec[k]); e->samp_graph_exec[k] = nullptr; }
        if (e->samp_graph[k]) { (void)hipGraphDestroy(e->samp_graph[k]); e->samp_graph[k] = nullptr; }
    }
}

// the plan that serves a step (or a run of steps) whose highest position is pos_last
// (ps[2]: the fused plan of the wide tier on one GPU from the second attention pass on -- its attention launch is the variant
//  whose passes are shared by helper blocks, nl_tp.h; the first pass keeps the launch compiled without them: 0.7 us per layer)
int plan_index(const nl_engine *e, int pos_last) {
    if (!(e->fused && pos_last < e->fused_max_pos)) return 0;
    return (e->attn_helpers && pos_last >= TP_PASS) ? 2 : 1;
}
nl_engine::PlanSet &pick_plan(nl_engine *e, int pos_last) { return e->ps[plan_index(e, pos_last)]; }

int launch_step(nl_engine *e, int pos) {
    nl_engine::PlanSet &S = pick_plan(e, pos);
    if (S.exec) {
        HIPCK(e, hipGraphLaunch(S.exec, e->stream));
        return NL_OK;
    }
    return run_plan_eager(e, S);
}

// One step through the call box: true = served (the done word carries the step's argmax in *id); false = this plan has no call
// graph and the caller copies, launches and synchronises as before.
bool call_step(nl_engine *e, int token, int pos, int stream, int hostout, int *id, int *rc) {
    nl_engine::PlanSet &S = pick_plan(e, pos);
    *rc = NL_OK;
    if (!S.call_exec) return false;
    const unsigned seq = ++e->box_seq;
    e->h_box[CTL_TOKEN] = token; e->h_box[CTL_POS] = pos; e->h_box[CTL_CHAIN] = 0; e->h_box[CTL_STEP] = 0;
    e->h_box[CTL_STREAM] = stream; e->h_box[CTL_HOSTOUT] = hostout;
    __atomic_store_n(e->h_box + 8, (int)seq, __ATOMIC_RELEASE);
    if (hipGraphLaunch(S.call_exec, e->stream) != hipSuccess) { *rc = e->fail(NL_ERR_HIP, "hipGraphLaunch (call graph): %s", hipGetErrorString(hipGetLastError())); return true; }
    // spin on the done word; a step that does not answer within two seconds is looked at through the stream instead
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long d;
    for (unsigned spins = 0;; spins++) {
        d = __atomic_load_n(e->h_done, __ATOMIC_ACQUIRE);
        if ((unsigned)(d >> 32) == seq) break;
        if ((spins & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            if (hipStreamSynchronize(e->stream) != hipSuccess) { *rc = e->fail(NL_ERR_HIP, "call graph: %s", hipGetErrorString(hipGetLastError())); return true; }
            d = __atomic_load_n(e->h_done, __ATOMIC_ACQUIRE);
            if ((unsigned)(d >> 32) != seq) { *rc = e->fail(NL_ERR_HIP, "call graph: the step finished without its done word"); return true; }
            break;
        }
    }
    if (id) *id = (int)(unsigned)d;
    return true;
}

// (re)build both plans and their graphs
int build_all(nl_engine *e) {
    for (auto &S : e->ps) destroy_graphs(S);
    e->graph_steps = 1;
    build_plan(e, e->ps[0].ops, false);
    if (e->fused) build_plan(e, e->ps[1].ops, true);
    if (e->fused && e->attn_helpers) build_plan(e, e->ps[2].ops, true, true);
    const bool has_coll = (e->G > 1 || e->force_tp_plan) && !e->p2p.on;
    if (e->use_graph && !(e->cfg.flags & NL_FLAG_LOCAL_GROUP)) {
        for (int k = 0; k < (e->fused ? (e->attn_helpers ? 3 : 2) : 1); k++) {
            int rc = capture_graph(e, e->ps[k]);
            if (rc && has_coll) {
                // RCCL inside a captured graph is not guaranteed on every RCCL build: fall back to eager launches
                (void)hipGetLastError();
                destroy_graphs(e->ps[k]);
                if (!getenv("NL_QUIET")) fprintf(stderr, "[nanollama_hip] graph capture with RCCL failed (%s); using eager launches\n", e->err.c_str());
            } else if (rc) {
                return rc;
            }
        }
    }
    return NL_OK;
}

int set_ctl(nl_engine *e, int token, int pos, int chain, int stream, int hostout = 0) {
    e->h_ctl[CTL_TOKEN] = token; e->h_ctl[CTL_POS] = pos; e->h_ctl[CTL_CHAIN] = chain;
    e->h_ctl[CTL_STEP] = 0; e->h_ctl[CTL_STREAM] = stream; e->h_ctl[CTL_HOSTOUT] = hostout;
    HIPCK(e, hipMemcpyAsync(e->ctl, e->h_ctl, CTL_WORDS * sizeof(int), hipMemcpyHostToDevice, e->stream));
    return NL_OK;
}

int check_step_args(nl_engine *e, int stream, int token, int pos) {
    if (!e->finalized) return e->fail(NL_ERR_STATE, "forward before nl_finalize");
    if (stream < 0 || stream >= e->cfg.max_streams) return e->fail(NL_ERR_INVALID, "stream %d out of range", stream);
    if (token < 0 || token >= e->cfg.vocab) return e->fail(NL_ERR_INVALID, "token %d out of range [0,%d)", token, e->cfg.vocab);
    if (pos < 0 || pos >= e->cfg.seq_len) return e->fail(NL_ERR_INVALID, "pos %d out of range [0,%d)", pos, e->cfg.seq_len);
    return NL_OK;
}

// Reset (go/model.go:623-631) zeroes both caches; here it only drops the stream's high-water mark.  A step at
// pos > mark (legal through the ABI) would attend over rows the reference reads as zeros, so exactly those rows
// [mark, pos) are cleared before the step: one strided memset per cache, only on that unusual call pattern.
int note_positions(nl_engine *e, int stream, int pos, int n, hipStream_t st = nullptr) {
    if (!st) st = e->stream;
    int &mark = e->hw[stream];
    if (pos > mark) {
        const size_t pitch = (size_t)e->cfg.seq_len * e->hd * 4, width = (size_t)(pos - mark) * e->hd * 4;
        const size_t height = (size_t)e->cfg.n_layers * e->KVs;
        const size_t off = (size_t)stream * e->kv_stream_stride + (size_t)mark * e->hd;
        HIPCK(e, hipMemset2DAsync(e->kcache + off, pitch, 0, width, height, st));
        HIPCK(e, hipMemset2DAsync(e->vcache + off, pitch, 0, width, height, st));
    }
    mark = std::max(mark, std::min(pos + n, e->cfg.seq_len));
    return NL_OK;
}

// after a synchronize: did any poll of the push all-reduce give up?
int p2p_check(nl_engine *e) {
    if (!e->p2p.on) return NL_OK;
    unsigned st = 0;
    HIPCK(e, hipMemcpy(&st, e->p2p.status, sizeof st, hipMemcpyDeviceToHost));
    if (st) return e->fail(NL_ERR_COMM, "push all-reduce timed out waiting for a peer (status %u): a rank is missing or stalled", st);
    return NL_OK;
}

// After a synchronize: did a cluster exchange of a fused launch (nl_block.h, nl_group.h) give up?  HIP promises nothing
// about which workgroups of a launch are resident together, so on a GPU shared with another tenant a member can wait
// for a peer that has not been dispatched.  Forward cannot fail in the reference (go/model.go:490): the flag is
// cleared, the fused plan is retired for this handle, and the caller redoes its whole call on the general plan (every
// buffer a step writes -- residual stream, K/V rows of its positions, logits, ids -- is rewritten by the redo).
// nl_last_error carries a one-time note while the call returns NL_OK.
bool take_fused_timeout(nl_engine *e) {
    if (!e->h_status || !*e->h_status) return false;
    const unsigned st = *e->h_status;
    *e->h_status = 0;
    (void)hipMemsetAsync(e->tick + 1, 0, sizeof(unsigned), e->stream);
    (void)hipStreamSynchronize(e->stream);
    e->fused = false;
    e->fused_retired = true;
    destroy_samp_graphs(e);
    e->fail(NL_OK, "warning: a fused-launch cluster exchange timed out (status %u); the step was redone on the general plan, "
                   "which this handle keeps from now on", st);
    if (!getenv("NL_QUIET")) fprintf(stderr, "[nanollama_hip] %s\n", e->err.c_str());
    return true;
}

// ---- persistent greedy decode of the smallest tier (nl_persist.h) -----------------------------------------------------------
void pd_free_raw(nl_engine *e) {
    for (auto &row : e->pd.raw)
        for (uint8_t *&p : row) if (p) { (void)hipFree(p); p = nullptr; }
    if (e->pd.lm_raw) { (void)hipFree(e->pd.lm_raw); e->pd.lm_raw = nullptr; }
}
void pd_free(nl_engine *e) {
    pd_free_raw(e);
    nl_engine::Persist &d = e->pd;
    void *bufs[] = {d.wimg, d.lmimg, d.simg, d.lmsimg, d.gx, d.gqkv, d.go, d.gxp, d.gh, d.gam, d.census, d.status, d.dbg, d.norms, d.gtok, d.gpart, d.embd_q8};
    for (void *b : bufs) if (b) (void)hipFree(b);
    if (d.h_status) (void)hipHostFree(d.h_status);
    if (d.h_mail) (void)hipHostFree(d.h_mail);
    d = nl_engine::Persist{};
}

// nl_finalize: pack the lane images (registers of every compute unit's role, LM-head slice for its LDS) from the raw tensors.
// Anything missing or unsuitable just leaves the path off: the launch plans serve every call.
int pd_build(nl_engine *e) {
    nl_engine::Persist &d = e->pd;
    const nl_config &c = e->cfg;
    bool ok = d.candidate && (e->embd_type == WT_Q8_0 || e->embd_type == WT_Q4_0 || e->embd_type == WT_Q5_0) && e->G == 1 && !e->force_tp_plan && e->num_cus == PD_GRID;
    for (int l = 0; ok && l < c.n_layers; l++) {
        for (int i = 0; i < 7; i++) ok = ok && d.raw[l][i];
        const nl_engine::Layer &L = e->layers[l];
        ok = ok && !L.bq && !L.bk && !L.bv && !L.bo;
    }
    if (!ok) { d.candidate = false; pd_free_raw(e); return NL_OK; }
    const size_t nw = (size_t)PD_GRID * PD_SLOTS * PD_UNITS * PD_THREADS, nlm = (size_t)PD_GRID * PD_ULM * PD_THREADS;
    HIPCK(e, hipMalloc((void **)&d.wimg, nw * 2 * sizeof(uint4)));
    HIPCK(e, hipMalloc((void **)&d.simg, nw * sizeof(unsigned short)));
    HIPCK(e, hipMalloc((void **)&d.lmimg, nlm * 2 * sizeof(uint4)));
    HIPCK(e, hipMalloc((void **)&d.lmsimg, nlm * sizeof(unsigned short)));
    e->bytes_weights += nw * 34 + nlm * 34;
    PdPackParams K{};
    for (int l = 0; l < c.n_layers; l++)
        for (int i = 0; i < 7; i++) { K.raw[l][i] = d.raw[l][i]; K.rtype[l][i] = d.rtype[l][i]; }
    K.lm_raw = d.lm_raw ? d.lm_raw : e->embd_raw;        // tied head: output.weight missing -> token_embd (go/model.go:195-201)
    K.lm_type = d.lm_raw ? d.lm_type : (unsigned char)e->embd_type;
    if (e->embd_type != WT_Q8_0) {
        const long long nblk = (long long)c.vocab * (c.dim / 32);
        HIPCK(e, hipMalloc((void **)&d.embd_q8, (size_t)nblk * 34));
        e->bytes_weights += (size_t)nblk * 34;
        hipLaunchKernelGGL(pd_embd_q8_kernel, dim3(2048), dim3(256), 0, e->stream, e->embd_raw, e->embd_type, nblk, d.embd_q8);
        HIPCK(e, hipGetLastError());
    }
    K.D = c.dim; K.I = c.interm; K.H = c.n_heads; K.V = c.vocab; K.L = c.n_layers; K.KV = c.n_kv_heads;
    K.wimg = d.wimg; K.simg = d.simg; K.lmimg = d.lmimg; K.lmsimg = d.lmsimg;
    hipLaunchKernelGGL(pd_pack_kernel, dim3(PD_GRID, PD_SLOTS * PD_UNITS + PD_ULM), dim3(PD_THREADS), 0, e->stream, K);
    HIPCK(e, hipGetLastError());
    HIPCK(e, hipStreamSynchronize(e->stream));
    pd_free_raw(e);
    const size_t L1 = (size_t)c.n_layers + 1;
    HIPCK(e, dalloc(&d.gx, L1 * c.dim, &e->bytes_state));
    HIPCK(e, dalloc(&d.go, L1 * c.dim, &e->bytes_state));
    HIPCK(e, dalloc(&d.gqkv, L1 * 3 * c.dim, &e->bytes_state));
    HIPCK(e, hipMemset(d.gqkv, 0, L1 * 3 * c.dim * 8));
    HIPCK(e, dalloc(&d.gxp, L1 * c.dim, &e->bytes_state));
    HIPCK(e, dalloc(&d.gh, L1 * c.interm, &e->bytes_state));
    HIPCK(e, dalloc(&d.gam, (size_t)PD_GRID, &e->bytes_state));
    HIPCK(e, dalloc(&d.gpart, (size_t)c.n_layers * c.n_heads * (PD_PARTS - 1) * 66, &e->bytes_state));
    HIPCK(e, hipMemset(d.gpart, 0, (size_t)c.n_layers * c.n_heads * (PD_PARTS - 1) * 66 * 8));
    HIPCK(e, hipMemset(d.gx, 0, L1 * c.dim * 8)); HIPCK(e, hipMemset(d.go, 0, L1 * c.dim * 8)); HIPCK(e, hipMemset(d.gxp, 0, L1 * c.dim * 8));
    HIPCK(e, hipMemset(d.gh, 0, L1 * c.interm * 8)); HIPCK(e, hipMemset(d.gam, 0, (size_t)PD_GRID * 16));
    HIPCK(e, dalloc(&d.norms, ((size_t)c.n_layers * 2 + 1) * c.dim, &e->bytes_state));
    for (int l = 0; l < c.n_layers; l++) {
        HIPCK(e, hipMemcpy(d.norms + (size_t)(2 * l) * c.dim, e->layers[l].attn_norm, (size_t)c.dim * 4, hipMemcpyDeviceToDevice));
        HIPCK(e, hipMemcpy(d.norms + (size_t)(2 * l + 1) * c.dim, e->layers[l].ffn_norm, (size_t)c.dim * 4, hipMemcpyDeviceToDevice));
    }
    HIPCK(e, hipMemcpy(d.norms + (size_t)(2 * c.n_layers) * c.dim, e->output_norm, (size_t)c.dim * 4, hipMemcpyDeviceToDevice));
    HIPCK(e, dalloc(&d.census, (size_t)32, &e->bytes_state));       // two launches' words, used alternately (each launch zeroes the other's)
    HIPCK(e, hipMemset(d.census, 0, 32 * sizeof(unsigned)));
    HIPCK(e, dalloc(&d.gtok, (size_t)2, &e->bytes_state));
    HIPCK(e, hipMemset(d.gtok, 0, 16));
    HIPCK(e, hipHostMalloc((void **)&d.h_mail, 16 * sizeof(pd_u64), hipHostMallocMapped | hipHostMallocCoherent));   // (coherent by request, not by HIP_HOST_COHERENT's default: both sides poll these words)
    memset(d.h_mail, 0, 16 * sizeof(pd_u64));
    if (hipHostGetDevicePointer((void **)&d.d_mail, d.h_mail, 0) != hipSuccess) { (void)hipGetLastError(); d.d_mail = nullptr; }
    HIPCK(e, dalloc(&d.status, (size_t)4, &e->bytes_state));
    HIPCK(e, hipMemset(d.status, 0, 16));
    HIPCK(e, dalloc(&d.dbg, (size_t)64, &e->bytes_state));
    HIPCK(e, hipMemset(d.dbg, 0, 64 * sizeof(long long)));
    HIPCK(e, hipHostMalloc((void **)&d.h_status, sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
    *d.h_status = 0;
    // (the instantiations pd_shape_ok admits)
    const void *kfn = c.dim == 576 ? (c.n_kv_heads == 9 ? reinterpret_cast<const void *>(pd_decode_kernel<18, 48, 9>) : reinterpret_cast<const void *>(pd_decode_kernel<18, 48, 3>))
                                   : (c.n_kv_heads == 4 ? reinterpret_cast<const void *>(pd_decode_kernel<8, 16, 4>) : reinterpret_cast<const void *>(pd_decode_kernel<8, 16, 2>));
    if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pd_lds_bytes()) != hipSuccess) {
        (void)hipGetLastError();
        d.candidate = false;
        return NL_OK;
    }
    if (const char *mp = getenv("NL_PERSIST_MAX_POS")) d.max_pos = std::max(0, std::min(PD_MAX_POS, atoi(mp)));   // knob (tests, tools)
    if (const char *sl = getenv("NL_PERSIST_SPIN_LIMIT")) d.spin_limit = atoi(sl);   // knob (tests): 0 makes every poll give up
    if (const char *ss = getenv("NL_PERSIST_SESSION")) d.session_on = atoi(ss) != 0;  // knob (A/B): 0 = one launch per call
    if (const char *iu = getenv("NL_PERSIST_IDLE_US")) d.idle_ticks = std::max(1LL, std::min(2000000LL, atoll(iu))) * 100;   // at most 2 s
    if (!d.d_mail) d.session_on = false;
    d.max_pos = std::min(d.max_pos, c.seq_len);
    d.tag_base = 0;
    d.ready = true;
    return NL_OK;
}

bool pd_usable(const nl_engine *e, int pos, int n) {
    return e->pd.ready && !e->pd.retired && n > 0 && pos + n <= e->pd.max_pos && !e->gamma_row && e->G == 1;
}

// One launch = n greedy tokens from (token, pos); ids land in e->ids.  The caller synchronises and then asks pd_take_timeout.
int pd_launch(nl_engine *e, int stream, int token, int pos, int n, float *host_logits = nullptr, bool session = false) {
    nl_engine::Persist &d = e->pd;
    const nl_config &c = e->cfg;
    if (d.tag_base > 0xf0000000u) {      // (tags never repeat inside the life of the areas: start over from clean ones)
        const size_t L1 = (size_t)c.n_layers + 1;
        HIPCK(e, hipMemsetAsync(d.gx, 0, L1 * c.dim * 8, e->stream)); HIPCK(e, hipMemsetAsync(d.go, 0, L1 * c.dim * 8, e->stream));
        HIPCK(e, hipMemsetAsync(d.gqkv, 0, L1 * 3 * c.dim * 8, e->stream));
        HIPCK(e, hipMemsetAsync(d.gxp, 0, L1 * c.dim * 8, e->stream)); HIPCK(e, hipMemsetAsync(d.gh, 0, L1 * c.interm * 8, e->stream));
        HIPCK(e, hipMemsetAsync(d.gam, 0, (size_t)PD_GRID * 16, e->stream));
        HIPCK(e, hipMemsetAsync(d.gpart, 0, (size_t)c.n_layers * c.n_heads * (PD_PARTS - 1) * 66 * 8, e->stream));
        d.tag_base = 0;
    }
    PdParams P{};
    P.census = d.census + (d.launch_no & 1u) * 16; P.census_next = d.census + ((d.launch_no + 1u) & 1u) * 16;
    d.launch_no++;
    P.session = session ? 1 : 0; P.idle_ticks = d.idle_ticks; P.mbox = d.d_mail; P.host_done = d.d_mail ? d.d_mail + 8 : nullptr; P.gtok = d.gtok;
    P.D = c.dim; P.I = c.interm; P.H = c.n_heads; P.V = c.vocab; P.L = c.n_layers; P.seq_len = c.seq_len; P.rope_conj = c.rope_conjugate;
    P.n_steps = n; P.token0 = token; P.pos0 = pos; P.spin_limit = d.spin_limit;
    P.eps = c.rms_eps; P.scale = (float)(1.0 / std::sqrt((double)e->hd));
    P.tag_base = d.tag_base;
    d.tag_base += (unsigned)n + 2u;
    P.wimg = d.wimg; P.simg = d.simg; P.lmimg = d.lmimg; P.lmsimg = d.lmsimg;
    P.norms = d.norms;
    P.embd_raw = e->pd.embd_q8 ? e->pd.embd_q8 : e->embd_raw;
    P.rope_cos = e->rope_cos; P.rope_sin = e->rope_sin;
    P.kcache = e->kcache + (long long)stream * e->kv_stream_stride; P.vcache = e->vcache + (long long)stream * e->kv_stream_stride;
    P.kv_layer_stride = e->kv_layer_stride;
    P.gx = d.gx; P.gqkv = d.gqkv; P.go = d.go; P.gxp = d.gxp; P.gh = d.gh; P.gam = d.gam; P.gpart = d.gpart;
    P.ids_out = e->ids; P.logits = e->logits; P.host_logits = host_logits;
    P.status = d.status; P.host_status = d.h_status; P.dbg = d.dbg;
    if (c.dim == 576 && c.n_kv_heads == 9) hipLaunchKernelGGL((pd_decode_kernel<18, 48, 9>), dim3(PD_GRID), dim3(PD_THREADS), pd_lds_bytes(), e->stream, P);
    else if (c.dim == 576) hipLaunchKernelGGL((pd_decode_kernel<18, 48, 3>), dim3(PD_GRID), dim3(PD_THREADS), pd_lds_bytes(), e->stream, P);
    else if (c.n_kv_heads == 4) hipLaunchKernelGGL((pd_decode_kernel<8, 16, 4>), dim3(PD_GRID), dim3(PD_THREADS), pd_lds_bytes(), e->stream, P);
    else hipLaunchKernelGGL((pd_decode_kernel<8, 16, 2>), dim3(PD_GRID), dim3(PD_THREADS), pd_lds_bytes(), e->stream, P);
    HIPCK(e, hipGetLastError());
    d.launches++; d.tokens += n;
    return NL_OK;
}

// after a synchronize: did the persistent launch give up (a poll ran out, or the census was not 8 x 32)?  The caller then
// redoes its chunk on the launch plans -- every buffer the chunk writes (K / V rows, ids, logits) is rewritten by the redo --
// and this handle keeps them from now on.
bool pd_take_timeout(nl_engine *e) {
    nl_engine::Persist &d = e->pd;
    if (!d.h_status) return false;
    // (test knob, read per call: the first n launches of the handle are treated as census misses although they placed well --
    //  the host side of the policy below is exercised without a crowded device)
    const char *fk = getenv("NL_PERSIST_FAKE_CENSUS_MISS");
    if (!*d.h_status && fk && d.launches > 0 && d.fake_misses < atoi(fk)) { d.fake_misses++; *d.h_status = 64u; }
    if (!*d.h_status) return false;
    const unsigned st = *d.h_status;
    *d.h_status = 0;
    (void)hipMemsetAsync(d.status, 0, sizeof(unsigned), e->stream);
    (void)hipStreamSynchronize(e->stream);
    // A hand-off poll that ran out retires the path at once.  A census that was not 8 x 32 is about WHERE the launch landed, not
    // about the exchange protocol: the chunk is redone on the launch plans as well, but the handle tries the persistent launch
    // again -- up to three misses (a server that lost its fast path to one crowded moment would keep the slow one for good).
    d.retired = !(st == 64u && ++d.census_misses < 3);
    // (what the census saw: workgroups per XCD and arrivals of the launch that gave up -- its words are still there, the next
    //  launch would have zeroed them)
    unsigned cen[16] = {0};
    char seen[160] = "";
    if ((st & 64u) && d.census && d.launch_no > 0 &&
        hipMemcpy(cen, d.census + ((d.launch_no - 1u) & 1u) * 16, sizeof(cen), hipMemcpyDeviceToHost) == hipSuccess)
        snprintf(seen, sizeof(seen), " [per XCD %u %u %u %u %u %u %u %u, %u of %d arrived, %d compute units]", cen[0], cen[1], cen[2], cen[3], cen[4],
                 cen[5], cen[6], cen[7], cen[8], PD_GRID, e->num_cus);
    e->fail(NL_OK, "warning: the persistent decode launch gave up (status %u: %s%s); the chunk was redone on the launch plans%s", st,
            (st & 64u) ? "its workgroups were not placed 32 per XCD" : "a hand-off poll timed out", seen,
            d.retired ? ", which this handle keeps from now on" : " (the next call tries the persistent launch again)");
    if (!getenv("NL_QUIET")) fprintf(stderr, "[nanollama_hip] %s\n", e->err.c_str());
    return true;
}

// ---- resident session: per-call Forward without a launch per call ---------------------------------------------------------
// End the session (if one is live): a quit in the mailbox, the doorman sets the status bit every poll of the launch watches,
// the launch drains.  Every entry point that enqueues work behind e->stream calls this first -- not for correctness (work
// behind the stream waits for the launch, which ends by itself idle_ticks after its last command) but so as not to wait.
// A resident launch fills every compute unit of its device: a launch of ANOTHER handle on that device (two models in one server)
// would find no place until the session idles out.  One live session per device is on record; whoever is about to queue work on
// the device posts a quit into the other handle's mailbox first (its owner finds the launch gone at its next call and starts
// another: the idle-out path).  Best effort: if the owner posts a token at the same moment the quit is lost and the newcomer
// waits out the idle limit, as before.
std::mutex g_sess_mu;
nl_engine *g_live_session[64] = {};
void pd_yield_device(nl_engine *e) {
    if (e->dev < 0 || e->dev >= 64) return;
    std::lock_guard<std::mutex> lk(g_sess_mu);
    nl_engine *o = g_live_session[e->dev];
    if (o && o != e && o->pd.h_mail) __atomic_store_n(o->pd.h_mail, (pd_u64)0xffffffffu, __ATOMIC_RELEASE);
}
void pd_session_record(nl_engine *e, bool live) {
    if (e->dev < 0 || e->dev >= 64) return;
    std::lock_guard<std::mutex> lk(g_sess_mu);
    if (live) g_live_session[e->dev] = e;
    else if (g_live_session[e->dev] == e) g_live_session[e->dev] = nullptr;
}

int pd_session_close(nl_engine *e) {
    nl_engine::Persist &d = e->pd;
    if (!d.live) { pd_yield_device(e); return NL_OK; }
    __atomic_store_n(d.h_mail, (pd_u64)0xffffffffu, __ATOMIC_RELEASE);
    HIPCK(e, hipStreamSynchronize(e->stream));
    d.live = false;
    pd_session_record(e, false);
    const unsigned st = *d.h_status;
    if (st & 256u) {                       // the clean end (quit or idle); every other bit is a poll that saw it
        *d.h_status = 0;
        HIPCK(e, hipMemsetAsync(d.status, 0, sizeof(unsigned), e->stream));
    } else if (st) {
        pd_take_timeout(e);
    }
    return NL_OK;
}

// One Forward at (token, pos) on the session: continue the resident launch when this call is its next step, otherwise start
// one.  *served = false leaves the call to the caller's other paths (the path is off, retired, or gave up just now -- the
// launch plans redo the step: every buffer it writes is rewritten).  id = the step's argmax; logits (want_logits) are in
// e->h_logits when the call returns.
int pd_session_step(nl_engine *e, int stream, int token, int pos, bool want_logits, int *id, bool *served) {
    nl_engine::Persist &d = e->pd;
    *served = false;
    if (!d.session_on || !pd_usable(e, pos, 1) || (want_logits && !e->d_h_logits)) return pd_session_close(e);
    int rc;
    if (d.live && (d.s_stream != stream || d.s_next_pos != pos || (want_logits && !d.s_logits) || e->hw[stream] < pos || *d.h_status))
        if ((rc = pd_session_close(e))) return rc;
    if (!pd_usable(e, pos, 1)) return NL_OK;         // (the close may have retired the path)
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = note_positions(e, stream, pos, 1))) return rc;
        if (!d.live) {
            const int n = std::min(d.max_pos - pos, e->ids_cap);
            __atomic_store_n(d.h_mail, (pd_u64)0, __ATOMIC_RELAXED);
            __atomic_store_n(d.h_mail + 8, (pd_u64)0, __ATOMIC_RELEASE);
            pd_yield_device(e);
            if ((rc = pd_launch(e, stream, token, pos, n, want_logits ? e->d_h_logits : nullptr, true))) return rc;
            d.tokens -= n;              // (pd_launch counted the whole session; steps are counted as they are served)
            pd_session_record(e, true);
            d.live = true; d.s_logits = want_logits; d.s_stream = stream; d.s_next_pos = pos; d.s_step = 0; d.s_nsteps = n;
            d.sessions++;
        } else {
            __atomic_store_n(d.h_mail, ((pd_u64)(unsigned)d.s_step << 32) | (unsigned)token, __ATOMIC_RELEASE);
        }
        const unsigned want = (unsigned)d.s_step + 1u;
        const auto t0 = std::chrono::steady_clock::now();
        pd_u64 v = 0;
        bool done = false;
        for (unsigned spins = 0;; spins++) {
            v = __atomic_load_n(d.h_mail + 8, __ATOMIC_ACQUIRE);
            if ((unsigned)(v >> 32) == want) { done = true; break; }
            if (__atomic_load_n(d.h_status, __ATOMIC_ACQUIRE)) break;
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) break;
            __builtin_ia32_pause();
        }
        if (!done) {       // the launch ended (idle just as the command arrived, or a poll gave up) -- or is stuck
            HIPCK(e, hipStreamSynchronize(e->stream));
            d.live = false;
            v = __atomic_load_n(d.h_mail + 8, __ATOMIC_ACQUIRE);
            done = (unsigned)(v >> 32) == want;
            const unsigned st = *d.h_status;
            if (st & 256u) {
                *d.h_status = 0;
                HIPCK(e, hipMemsetAsync(d.status, 0, sizeof(unsigned), e->stream));
                if (!done) continue;                 // this call's step never began: a fresh session takes it
            } else if (!done || st) {
                if (!st) *d.h_status = 1024u;        // (no word from the launch at all)
                pd_take_timeout(e);
                return NL_OK;
            }
        }
        if (id) *id = (int)(unsigned)v;
        d.s_step++; d.s_next_pos++; d.tokens++;
        if (d.s_step == d.s_nsteps) d.live = false;   // (its last step: the launch leaves by itself)
        *served = true;
        return NL_OK;
    }
    return NL_OK;
}

struct Slot { int layer; std::string field; };

bool parse_name(const char *name, Slot &s) {
    int li = -1, off = 0;
    if (sscanf(name, "blk.%d.%n", &li, &off) >= 1 && off > 0) { s.layer = li; s.field = name + off; return true; }
    s.layer = -1; s.field = name;
    return true;
}

}  // namespace

namespace {
// x[n][ldx] (f32) -> the fp16 hi/lo fragment store the MFMA kernel DMAs into LDS (layout: nl_qgemm.h)
hipError_t launch_xsplit(int wtype, const float *x, int ldx, int cols, int n_tokens, uint4 *xf, hipStream_t st) {
    const int nt16 = ((n_tokens + 63) / 64) * 4;
    const long long total = (long long)(cols / 32) * nt16 * 64;
    hipLaunchKernelGGL(xsplit_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)), dim3(256), 0, st,
                       x, ldx, cols / 32, n_tokens, nt16, wtype == WT_Q4_0 ? 1 : 0, xf);
    return hipGetLastError();
}

// Long token runs over Q4_0 weights take the weights-through-LDS kernel (nl_qgemm2.h: 1.2-1.5x qgemm_kernel from 128
// tokens up, tools/qgemm2_bench.hip; Q8_0's 32-byte rows cost more to expand and stay on qgemm_kernel).  Workgroups of
// 64 rows: x 256 tokens (two token tiles per wavefront) when that still gives every compute unit two of them, else x 64.
bool qgemm2_ok(int wtype, int n_tokens) {
    const char *knob = getenv("NL_QG2_MIN_TOKENS");   // knob (tests, tools; read per launch so a test can flip it): a huge value disables
    const int min_n = knob ? atoi(knob) : 128;
    return wtype == WT_Q4_0 && n_tokens >= min_n;
}
// the plain GEMM of a long Q4_0 token run whose unsplit 64-row x 64-token grid fills the chip: qgemm2_kernel, no split-K
bool qgemm2_plain_unsplit(int wtype, int ntiles, int n_tokens) {
    return qgemm2_ok(wtype, n_tokens) && ((ntiles + 3) / 4) * ((n_tokens + QG_TOK - 1) / QG_TOK) >= 128;
}
template <int EPI>
hipError_t launch_qgemm2(const QGemmParams &P, int row_blocks, hipStream_t st) {
    if ((long long)row_blocks * ((P.n_tokens + 255) / 256) >= 512)
        hipLaunchKernelGGL((qgemm2_kernel<WT_Q4_0, 4, 2, 8, EPI>), dim3(row_blocks, (P.n_tokens + 255) / 256, 1), dim3(512), 0, st, P);
    else
        hipLaunchKernelGGL((qgemm2_kernel<WT_Q4_0, 4, 1, 4, EPI>), dim3(row_blocks, (P.n_tokens + 63) / 64, 1), dim3(256), 0, st, P);
    return hipGetLastError();
}

// gate || up with the SwiGLU epilogue (nl_qgemm.h): 4 wavefronts x (gate tile, up tile) per workgroup, no split-K.
// Worth it only when the unsplit grid fills the chip; the caller falls back to the plain launch + bswiglu otherwise.
constexpr int QG_FUSED_WAVES = 4;
bool qgemm_swiglu_fits(int ntiles, int n_tokens) {
    static const int min_wg = getenv("NL_FUSED_SWIGLU_MIN_WG") ? atoi(getenv("NL_FUSED_SWIGLU_MIN_WG")) : 128;   // knob: tests force 1, tools disable with a huge value
    // (a step of <= 32 tokens runs the plain kernel on one / two 16-token tiles instead: big x 8 streams 7.76 -> 7.55 ms)
    return (n_tokens > 32 || min_wg <= 1) && ((ntiles + QG_FUSED_WAVES - 1) / QG_FUSED_WAVES) * ((n_tokens + QG_TOK - 1) / QG_TOK) >= min_wg;
}
hipError_t launch_qgemm_swiglu(int wtype, QGemmParams P, hipStream_t st) {
    P.nt16 = ((P.n_tokens + 63) / 64) * 4;
    P.ksplit = 1;
    if (qgemm2_ok(wtype, P.n_tokens)) return launch_qgemm2<QG_EPI_SWIGLU>(P, (P.ntiles + 1) / 2, st);
    const dim3 grid((P.ntiles + QG_FUSED_WAVES - 1) / QG_FUSED_WAVES, (P.n_tokens + QG_TOK - 1) / QG_TOK, 1);
    switch (wtype) {
    case WT_Q4_0: hipLaunchKernelGGL((qgemm_kernel<WT_Q4_0, QG_FUSED_WAVES, 2, QG_EPI_SWIGLU>), grid, dim3(QG_FUSED_WAVES * 64), 0, st, P); break;
    case WT_Q8_0: hipLaunchKernelGGL((qgemm_kernel<WT_Q8_0, QG_FUSED_WAVES, 2, QG_EPI_SWIGLU>), grid, dim3(QG_FUSED_WAVES * 64), 0, st, P); break;
    case WT_F16: hipLaunchKernelGGL((qgemm_kernel<WT_F16, QG_FUSED_WAVES, 2, QG_EPI_SWIGLU>), grid, dim3(QG_FUSED_WAVES * 64), 0, st, P); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Q|K|V GEMM with the RoPE + KV-store epilogue (nl_qgemm.h): 128-row workgroups, no split-K -- for steps whose
// unsplit grid fills the chip (prompts); decode batches keep the split-K GEMM + brope_kv.
bool qgemm_rope_fits(int ntiles, int n_tokens) {
    static const int min_wg = getenv("NL_FUSED_ROPE_MIN_WG") ? atoi(getenv("NL_FUSED_ROPE_MIN_WG")) : 128;   // knob: tests force 1
    return ((ntiles + QG_WAVES - 1) / QG_WAVES) * ((n_tokens + QG_TOK - 1) / QG_TOK) >= min_wg;
}
hipError_t launch_qgemm_rope(int wtype, QGemmParams P, hipStream_t st) {
    P.nt16 = ((P.n_tokens + 63) / 64) * 4;
    P.ksplit = 1;
    if (qgemm2_ok(wtype, P.n_tokens)) return launch_qgemm2<QG_EPI_ROPE>(P, (P.ntiles + 3) / 4, st);
    P.row_groups = (P.ntiles + QG_WAVES - 1) / QG_WAVES;
    const dim3 grid(P.row_groups, (P.n_tokens + QG_TOK - 1) / QG_TOK, 1);
    switch (wtype) {
    case WT_Q4_0: hipLaunchKernelGGL((qgemm_kernel<WT_Q4_0, QG_WAVES, 1, QG_EPI_ROPE>), grid, dim3(QG_WAVES * 64), 0, st, P); break;
    case WT_Q8_0: hipLaunchKernelGGL((qgemm_kernel<WT_Q8_0, QG_WAVES, 1, QG_EPI_ROPE>), grid, dim3(QG_WAVES * 64), 0, st, P); break;
    case WT_F16: hipLaunchKernelGGL((qgemm_kernel<WT_F16, QG_WAVES, 1, QG_EPI_ROPE>), grid, dim3(QG_WAVES * 64), 0, st, P); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int WT, int WAVES, int RT = QG_RT>
void launch_qgemm_nt(int nt, dim3 grid, hipStream_t st, const QGemmParams &P) {
    // one 64-token tile (decode batches, short prompts) in 64-row workgroups: fragments staged through registers (RSTAGE)
    static const bool rstage_knob = !(getenv("NL_QG_RSTAGE") && atoi(getenv("NL_QG_RSTAGE")) == 0);   // developer knob (tools/)
    if (WAVES == 4 && RT == 1 && grid.y == 1 && rstage_knob) {
        switch (nt) {
        case 1: hipLaunchKernelGGL((qgemm_kernel<WT, 4, 1, QG_EPI_PLAIN, 1, true>), grid, dim3(256), 0, st, P); break;
        case 2: hipLaunchKernelGGL((qgemm_kernel<WT, 4, 1, QG_EPI_PLAIN, 2, true>), grid, dim3(256), 0, st, P); break;
        default: hipLaunchKernelGGL((qgemm_kernel<WT, 4, 1, QG_EPI_PLAIN, 4, true>), grid, dim3(256), 0, st, P); break;
        }
        return;
    }
    switch (nt) {
    case 1: hipLaunchKernelGGL((qgemm_kernel<WT, WAVES, RT, QG_EPI_PLAIN, 1>), grid, dim3(WAVES * 64), 0, st, P); break;
    case 2: hipLaunchKernelGGL((qgemm_kernel<WT, WAVES, RT, QG_EPI_PLAIN, 2>), grid, dim3(WAVES * 64), 0, st, P); break;
    default: hipLaunchKernelGGL((qgemm_kernel<WT, WAVES, RT, QG_EPI_PLAIN, 4>), grid, dim3(WAVES * 64), 0, st, P); break;
    }
}

hipError_t launch_qgemm(int wtype, QGemmParams P, hipStream_t st, float *part_buf = nullptr, size_t part_cap = 0,
                        int *ks_out = nullptr) {
    P.nt16 = ((P.n_tokens + 63) / 64) * 4;
    const int tok_tiles = (P.n_tokens + QG_TOK - 1) / QG_TOK;
    const int nchunks = (P.cols / 32 + QG_KC - 1) / QG_KC;
    const int mats = P.q1 ? 2 : 1;
    if (qgemm2_plain_unsplit(wtype, P.ntiles, P.n_tokens) && mats == 1) {   // unsplit grid fills the chip
        P.ksplit = 1;
        P.part = nullptr;
        if (ks_out) *ks_out = 1;
        return launch_qgemm2<QG_EPI_PLAIN>(P, (P.ntiles + 3) / 4, st);
    }
    if (P.nrm_out.w) return hipErrorInvalidValue;   // (the folded norm's producer epilogue exists in qgemm2_kernel only: host logic error)
    // Workgroup height: 128 rows (8 wavefronts) when that alone fills the chip, else 64 rows (4 wavefronts) --
    // twice the workgroups and half the split-K for the small-N decode batches (goldie shapes at 16-128 tokens:
    // -4...-18 % per launch, tools/qgemm_variants.sh); then split K until ~128 workgroups exist.
    static const int force_waves = getenv("NL_QG_FORCE_WAVES") ? atoi(getenv("NL_QG_FORCE_WAVES")) : 0;   // developer knob
    int waves = QG_WAVES, rt = QG_RT;
    if (((P.ntiles + QG_WAVES - 1) / QG_WAVES) * mats * tok_tiles < 128) waves = 4;
    if (force_waves == 4 || force_waves == 8) waves = force_waves;
    // decode batches (one 64-token tile): a workgroup's ingest is dominated by the activation fragments (256 B per
    // column against 72 B of Q4_0 weights for a 128-row workgroup), so tall workgroups -- 8 wavefronts x 2 tiles = 256
    // rows -- with a deep K split move the least data per launch
    static const int dec_geom = getenv("NL_QG_DEC_GEOM") ? atoi(getenv("NL_QG_DEC_GEOM")) : 0;   // developer knob: 82 = 8 waves x 2 tiles
    if (tok_tiles == 1 && part_buf && P.ldo == P.rows && dec_geom == 82 && QG_RT == 1) { waves = 8; rt = 2; }
    const int row_groups = (P.ntiles + waves * rt - 1) / (waves * rt);
    P.row_groups = row_groups;
    int ks = 1;
    if (part_buf && P.ldo == P.rows) {
        static const int ks_cap = getenv("NL_KS_CAP") ? atoi(getenv("NL_KS_CAP")) : 16;   // developer knob (tools/)
        static const int min_wg = getenv("NL_QG_MIN_WG") ? atoi(getenv("NL_QG_MIN_WG")) : 128;
        static const int max_chunks = getenv("NL_QG_MAX_CHUNKS") ? atoi(getenv("NL_QG_MAX_CHUNKS")) : 6;   // goldie x 64 streams: 2.06 -> 1.94 ms per step (tools/sweep_qg_split.sh)
        static const int max_wg = getenv("NL_QG_MAX_WG") ? atoi(getenv("NL_QG_MAX_WG")) : 1024;
        // split K until the grid fills the chip; a decode batch (one token tile) keeps splitting while a workgroup would
        // still walk more than max_chunks 128-column chunks one after the other (each is a dependent memory round trip)
        while (ks * 2 <= nchunks && ks < ks_cap &&
               (row_groups * mats * tok_tiles * ks < min_wg ||
                (tok_tiles == 1 && (nchunks + ks - 1) / ks > max_chunks && row_groups * mats * ks * 2 <= max_wg)))
            ks *= 2;
        while (ks > 1 && (size_t)ks * P.n_tokens * P.ldo > part_cap) ks /= 2;
    }
    P.ksplit = ks;
    P.part = part_buf;
    dim3 grid(row_groups * mats, tok_tiles, ks);
    // a step of <= 16 / 32 tokens fetches and multiplies only the first one / two 16-token tiles of its group
    const int nt = tok_tiles > 1 || P.n_tokens > 32 ? 4 : P.n_tokens > 16 ? 2 : 1;
    if (wtype != WT_Q4_0 && wtype != WT_Q8_0 && wtype != WT_F16) return hipErrorInvalidValue;
    if (rt == 2 && waves == 8) {
        if (wtype == WT_Q4_0) launch_qgemm_nt<WT_Q4_0, 8, 2>(nt, grid, st, P);
        else if (wtype == WT_Q8_0) launch_qgemm_nt<WT_Q8_0, 8, 2>(nt, grid, st, P);
        else launch_qgemm_nt<WT_F16, 8, 2>(nt, grid, st, P);
    } else if (waves == 4) {
        if (wtype == WT_Q4_0) launch_qgemm_nt<WT_Q4_0, 4>(nt, grid, st, P);
        else if (wtype == WT_Q8_0) launch_qgemm_nt<WT_Q8_0, 4>(nt, grid, st, P);
        else launch_qgemm_nt<WT_F16, 4>(nt, grid, st, P);
    } else {
        if (wtype == WT_Q4_0) launch_qgemm_nt<WT_Q4_0, QG_WAVES>(nt, grid, st, P);
        else if (wtype == WT_Q8_0) launch_qgemm_nt<WT_Q8_0, QG_WAVES>(nt, grid, st, P);
        else launch_qgemm_nt<WT_F16, QG_WAVES>(nt, grid, st, P);
    }
    hipError_t s = hipGetLastError();
    if (ks_out) { *ks_out = ks; return s; }     // the consumer kernel adds the slabs (GemmOut, nl_batch.h)
    if (s != hipSuccess || ks == 1) return s;
    const long long count = (long long)P.n_tokens * P.ldo;
    hipLaunchKernelGGL(qgemm_sum_kernel, dim3((unsigned)std::min<long long>((count + 255) / 256, 2048)), dim3(256), 0, st,
                       part_buf, ks, count, P.resid, P.out, P.bias, P.ldo);
    return hipGetLastError();
}

// ---- short token runs on dgemm_kernel (nl_dgemm.h): Q4_0, whole 256-column groups, 32-row producer blocks ----
bool dgemm_mat_ok(const PackedMat &m, bool rows32, int T, bool ssq) {
    return m.wtype == WT_Q4_0 && dg_cols_ok(m.cols, T, ssq) && (!rows32 || m.rows % 32 == 0) && m.ntiles * TR == m.rows;
}
hipError_t launch_dgemm_rope(const QGemmParams &P, hipStream_t st) { return dg_launch_rope(P, st); }
hipError_t launch_dgemm_swiglu(const QGemmParams &P, hipStream_t st) { return dg_launch_swiglu(P, st); }
hipError_t launch_dgemm_plain(const QGemmParams &P, hipStream_t st) { return dg_launch_plain(P, st); }
hipError_t launch_dgemm_head(const QGemmParams &P, hipStream_t st, int num_cus) { return dg_launch_head(P, st, num_cus); }
}  // namespace

namespace {

// below this many tokens the ~10 launches per layer of the multi-token step cost more than n single-token steps
// (measured: 2 streams nano 0.61 ms batched vs 0.55 ms as two single steps, goldie 1.77 vs 1.47; 3 streams 0.61 vs
// 0.82 and 1.77 vs 2.19)
const int NL_BATCH_MIN = getenv("NL_BATCH_MIN") ? atoi(getenv("NL_BATCH_MIN")) : 3;   // env: developer knob (tools/)

bool batch_supported(const nl_engine *e) {
    if (e->G != 1 || e->force_tp_plan) return false;
    auto ok = [](const PackedMat &m) { return m.wtype == WT_Q4_0 || m.wtype == WT_Q8_0 || m.wtype == WT_F16; };
    for (const auto &L : e->layers)
        if (!ok(L.qkv) || !ok(L.wo) || !ok(L.gate) || !ok(L.up) || !ok(L.down)) return false;
    return ok(e->lm_head) && !getenv("NL_NO_BATCH_PATH");
}

int batch_alloc(nl_engine *e, nl_engine::Batch &b, int cap_limit = 2048) {
    if (b.ready) return NL_OK;
    const nl_config &c = e->cfg;
    // tokens per multi-token step: a whole prompt when it fits (one launch per op and layer, enough workgroups
    // that no GEMM needs split-K); halved until the activation set stays under 4 GiB.  The LM head runs on
    // <= 64 of them.
    const size_t R = (size_t)(e->Hs + 2 * e->KVs) * e->hd, HQ = (size_t)e->Hs * e->hd;
    b.cap = std::min(cap_limit, ((c.seq_len + QG_TOK - 1) / QG_TOK) * QG_TOK);
    {
        const size_t per_tok = 4 * (2 * (size_t)c.dim + R + 2 * HQ + 3 * (size_t)e->Is + (size_t)e->Hs * e->nsplit_max * (e->hd + 2) +
                                    std::max<size_t>(std::max<size_t>(HQ, c.dim), e->Is));
        while (b.cap > 512 && per_tok * b.cap > ((size_t)4 << 30)) b.cap /= 2;
    }
    b.lm_cap = QG_TOK;
    const size_t n = b.cap;
    HIPCK(e, dalloc(&b.x, n * c.dim, &e->bytes_state));
    HIPCK(e, dalloc(&b.qkv, n * R, &e->bytes_state));
    HIPCK(e, dalloc(&b.q, n * HQ, &e->bytes_state));
    HIPCK(e, dalloc(&b.g, n * e->Is, &e->bytes_state));
    HIPCK(e, dalloc(&b.u, n * e->Is, &e->bytes_state));
    HIPCK(e, dalloc(&b.logits, (size_t)b.lm_cap * c.vocab, &e->bytes_state));
    {
        const size_t nx = xfrag_uint4((int)std::max<size_t>(std::max<size_t>(HQ, c.dim), e->Is), (int)n);
        float *raw = nullptr;
        HIPCK(e, dalloc(&raw, nx * 4, &e->bytes_state));
        b.xfrag = reinterpret_cast<uint4 *>(raw);
        const size_t nx2 = xfrag_uint4(std::max(e->Is, c.dim), (int)n);   // SiLU(gate) * up, written by the fused gate/up GEMM (or the folded norm's fragments)
        raw = nullptr;
        HIPCK(e, dalloc(&raw, nx2 * 4, &e->bytes_state));
        b.xfrag2 = reinterpret_cast<uint4 *>(raw);
    }
    HIPCK(e, dalloc(&b.ssq, n * (size_t)((c.dim + 31) / 32), &e->bytes_state));   // (one partial per 64 rows on qgemm2_kernel, per 32 on dgemm_kernel)
    HIPCK(e, dalloc(&b.nscale, 2 * n, &e->bytes_state));
    b.kpart_cap = (size_t)16 * QG_TOK * std::max<size_t>(std::max<size_t>(R, c.dim), e->Is);
    HIPCK(e, dalloc(&b.kpart, b.kpart_cap, &e->bytes_state));
    HIPCK(e, dalloc(&b.kpart2, b.kpart_cap, &e->bytes_state));
    HIPCK(e, dalloc(&b.part_o, n * e->Hs * e->nsplit_max * e->hd, &e->bytes_state));
    HIPCK(e, dalloc(&b.part_ml, n * e->Hs * e->nsplit_max * 2, &e->bytes_state));
    HIPCK(e, dalloc(&b.tok, 5 * n, &e->bytes_state));   // token | pos | stream | attention workgroup list | partials per token, the layout of h_meta: one upload per step
    b.pos = b.tok + n;
    b.stream = b.tok + 2 * n;
    HIPCK(e, dalloc(&b.ids, n, &e->bytes_state));
    HIPCK(e, dalloc(&b.tcos, n * (size_t)(e->hd / 2), &e->bytes_state));
    HIPCK(e, dalloc(&b.tsin, n * (size_t)(e->hd / 2), &e->bytes_state));
    HIPCK(e, dalloc(&b.tkv, n, &e->bytes_state));
    {   // one layer's K / V^T as fp16 hi / lo LDS images (Kv16Image), rebuilt per layer of a prompt step
        const size_t img = e->hd == 64 ? Kv16Image<64>::BYTES : Kv16Image<32>::BYTES;
        HIPCK(e, dalloc(&b.kv16, (size_t)e->KVs * e->nsplit_max * img / 16, &e->bytes_state));
    }
    HIPCK(e, hipHostMalloc((void **)&b.h_meta, 5 * n * sizeof(int), hipHostMallocDefault));
    b.ready = true;
    return NL_OK;
}

// a short multi-token step (dgemm_step_ok: decode batches, prompts of up to a few hundred tokens) of a model whose layer matrices are all Q4_0 with whole 256-column groups
// runs its five GEMM-shaped launches per layer on dgemm_kernel (nl_dgemm.h) with the RMSNorms folded around them
bool dgemm_model_ok(const nl_engine *e) {
    const nl_config &c = e->cfg;
    if (c.qk_norm || c.dim % 32 || c.n_layers <= 0 || c.dim / 32 > DG_SSQ_MAX_NRB) return false;   // (nrb: the consumer's LDS area for the partial sums of squares)
    for (const auto &L : e->layers)
        if (!dgemm_mat_ok(L.qkv, false, 3, true) || !dgemm_mat_ok(L.wo, true, 2, false) || !dgemm_mat_ok(L.gate, true, 8, true) ||
            !dgemm_mat_ok(L.up, true, 8, true) || !dgemm_mat_ok(L.down, true, 2, false) || L.gate.rows != L.up.rows || L.gate.rows % 64 ||
            L.wo.rows != c.dim || L.down.rows != c.dim)
            return false;
    return true;
}
// the block-major weight copies dgemm_kernel reads, once per handle, at its first multi-token step (outside any capture; the
// caller holds g_setup_mu)
int dgemm_prepare(nl_engine *e) {
    if (e->dg_state) return NL_OK;
    e->dg_state = -1;
    if (getenv("NL_DGEMM") && atoi(getenv("NL_DGEMM")) == 0) { e->dg_state = 0; return NL_OK; }     // (knob: looked at again next step)
    if (!batch_supported(e) || !dgemm_model_ok(e)) return NL_OK;
    for (auto &L : e->layers)
        for (PackedMat *m : {&L.qkv, &L.wo, &L.gate, &L.up, &L.down}) {
            HIPCK(e, arena_alloc(e, (void **)&m->q3, m->q_bytes));
            HIPCK(e, arena_alloc(e, (void **)&m->s3, m->s_bytes));
            e->bytes_weights += m->q_bytes + m->s_bytes;
            const long long groups = (long long)m->ntiles * (m->npairs / KL);
            hipLaunchKernelGGL(dg_permute_kernel, dim3((unsigned)std::min<long long>((groups * 128 + 255) / 256, 65535)), dim3(256), 0, e->stream,
                               reinterpret_cast<const uint4 *>(m->q), m->s, reinterpret_cast<uint4 *>(m->q3), m->s3, groups);
            HIPCK(e, hipGetLastError());
        }
    // the LM head of a decode batch: dghead_kernel (resident workgroups, the final RMSNorm folded into the last down launch)
    e->dg_head = false;
    if (e->lm_head.wtype == WT_Q4_0 && dg_head_ok(e->lm_head.rows, e->lm_head.cols) && e->lm_head.ntiles * TR == e->lm_head.rows &&
        e->lm_head.rows == e->cfg.vocab && e->lm_head.cols == e->cfg.dim) {
        PackedMat *m = &e->lm_head;
        HIPCK(e, arena_alloc(e, (void **)&m->q3, m->q_bytes));
        HIPCK(e, arena_alloc(e, (void **)&m->s3, m->s_bytes));
        e->bytes_weights += m->q_bytes + m->s_bytes;
        const long long groups = (long long)m->ntiles * (m->npairs / KL);
        hipLaunchKernelGGL(dg_permute_kernel, dim3((unsigned)std::min<long long>((groups * 128 + 255) / 256, 65535)), dim3(256), 0, e->stream,
                           reinterpret_cast<const uint4 *>(m->q), m->s, reinterpret_cast<uint4 *>(m->q3), m->s3, groups);
        HIPCK(e, hipGetLastError());
        e->dg_head = true;
    }
    HIPCK(e, hipStreamSynchronize(e->stream));
    e->dg_state = 1;
    return NL_OK;
}
bool dgemm_step_ok(const nl_engine *e, int n) {
    if (e->dg_state != 1) return false;
    const char *k = getenv("NL_DGEMM");                 // knob (tests, tools; read per step): 0 keeps the split-K launches
    if (k && atoi(k) == 0) return false;
    const char *mk = getenv("NL_DGEMM_MAX_TOKENS");
    if (mk) return n <= atoi(mk);
    // Decode batches and prompts while the largest launch (gate || up: ceil(n / 16) token tiles x I / 64 row groups) stays within
    // about four rounds of workgroups: measured ahead of the long-run GEMM of nl_qgemm2.h up to 256 tokens on goldie (768 / 1024
    // workgroups: 2.77 against 2.88 ms, 3.23 against 3.21) and 512 on mini (1.60 against 1.74 ms; 768 tokens 2.21 against 2.19) --
    // tools/bench_short_prompt.py
    return (long long)((n + 15) / 16) * (e->cfg.interm / 64) <= 1024;
}

// GEMM of the multi-token step: input = the fragment store the producing kernel just filled; output = `out`
// (with resid / bias applied) when it ran unsplit, else split-K slabs in `part` for the consumer to add.
hipError_t qg(nl_engine *e, nl_engine::Batch &bt, const PackedMat &m, int n, float *out, int ldo, const float *resid, hipStream_t st,
              GemmOut *res, float *part, const float *bias = nullptr, const uint4 *xf = nullptr,
              const QGemmParams::NormOut *nout = nullptr, int x1 = 0) {
    QGemmParams P{};
    P.bias = bias;
    P.x1 = x1;
    if (nout) P.nrm_out = *nout;
    P.q = m.q; P.s = m.s; P.rows = m.rows; P.cols = m.cols; P.npairs = m.npairs; P.ntiles = m.ntiles;
    P.xf = xf ? xf : bt.xfrag; P.n_tokens = n; P.out = out; P.ldo = ldo; P.resid = resid;
    int ks = 1;
    hipError_t s = launch_qgemm(m.wtype, P, st, part, bt.kpart_cap, &ks);
    *res = GemmOut{out, part, ks, (long long)n * ldo, bias};
    return s;
}

hipError_t launch_qgemm_plain(nl_engine *e, nl_engine::Batch &bt, const PackedMat &m, int n, float *out, int ldo, hipStream_t st) {
    QGemmParams P{};
    P.q = m.q; P.s = m.s; P.rows = m.rows; P.cols = m.cols; P.npairs = m.npairs; P.ntiles = m.ntiles;
    P.xf = bt.xfrag; P.n_tokens = n; P.out = out; P.ldo = ldo;
    return launch_qgemm(m.wtype, P, st, bt.kpart, bt.kpart_cap);   // split-K + its own sum launch
}

// One multi-token step: n <= 64 (token, pos, stream) triples through every layer on the MFMA path.
// lm_mode: 0 = no LM head, 1 = logits + argmax for every token, 2 = logits + argmax for the LAST token only.
// The caller has filled bt.h_meta; stream-ordered, no synchronisation inside.
int batched_step(nl_engine *e, nl_engine::Batch &b, hipStream_t st, int n, int lm_mode, bool one_stream = false) {
    const nl_config &c = e->cfg;
    const int D = c.dim, hd = e->hd, HQ = e->Hs * hd, R = (e->Hs + 2 * e->KVs) * hd;
#define LCK(expr) do { hipError_t s_ = (expr); if (s_ != hipSuccess) return e->fail(NL_ERR_HIP, "batched step: %s: %s", #expr, hipGetErrorString(s_)); } while (0)
    // prompts (consecutive positions of one stream): the workgroups of the causal attention launch, each a query tile and a
    // run of consecutive 128-key chunks under the diagonal (AttnParams::live_map), and how many partials every token gets.
    // Run length: the shortest that lets every workgroup be resident at once (2 per CU); tiles are cut into equal runs.
    // Prompts of several query tiles (n >= NL_KV16_MIN_TOKENS): the chunks are split into fp16 halves once per layer
    // (kv16_build_kernel) and a workgroup takes twice the rows, one workgroup per CU (AttnTile16ShadowQT).
    int n_live = 0, max_nparts = 0;
    bool kv16_on = false;
    {
        bool cons = one_stream && n >= 8 && attn_tile_supported(e->gqa) && !attn_knobs().f32_tile && !attn_knobs().no_tile;
        for (int i = 1; i < n && cons; i++) cons = b.h_meta[b.cap + i] == b.h_meta[b.cap] + i;
        const int kv16_min = getenv("NL_KV16_MIN_TOKENS") ? atoi(getenv("NL_KV16_MIN_TOKENS")) : 256;   // (read per step: a test lowers it)
        kv16_on = cons && n >= kv16_min && b.kv16 != nullptr;
        const int qt = attn_tile16_qt(e->gqa) * (kv16_on ? 2 : 1), ntile = (n + qt - 1) / qt, p0 = b.h_meta[b.cap];
        const long long slots = (long long)e->num_cus * (kv16_on ? 1 : 2);   // workgroups resident at once
        if (cons && ntile <= 256 && e->KVs <= 255) {
            auto chunks_of = [&](int z) { return std::min((p0 + std::min((z + 1) * qt, n) - 1) / ATT_CH + 1, e->nsplit_max); };
            const int run_knob = getenv("NL_ATT_RUN") ? atoi(getenv("NL_ATT_RUN")) : 0;   // developer knob: fixed run length (read per step: a test sets it)
            int run = run_knob > 0 ? run_knob : 1;
            for (; run_knob <= 0 && run < e->nsplit_max; run++) {
                long long wgs = 0;
                for (int z = 0; z < ntile; z++) wgs += (chunks_of(z) + run - 1) / run;
                if (wgs * e->KVs <= slots) break;
            }
            int *map = b.h_meta + 3 * b.cap, *nparts = b.h_meta + 4 * b.cap;
            std::vector<int> wg;
            for (int z = ntile - 1; z >= 0; z--) {
                const int cz = chunks_of(z), nrun = (cz + run - 1) / run, len = (cz + nrun - 1) / nrun;
                for (int k = 0; k * len < cz; k++)
                    for (int kv = 0; kv < e->KVs; kv++)
                        wg.push_back((int)((unsigned)kv << 24) | (z << 16) | (k << 12) | ((k * len) << 6) | std::min(len, cz - k * len));
                for (int i = z * qt; i < std::min((z + 1) * qt, n); i++) nparts[i] = ((p0 + i) / ATT_CH) / len + 1;
                max_nparts = std::max(max_nparts, nrun);
            }
            if ((int)wg.size() <= b.cap) {
                // the longest runs first; with two workgroups per CU, i and i + #CUs share one (tools/att_stamps.py census):
                // the entries past the first #CUs follow shortest first, so every CU's pair does about the same work
                std::stable_sort(wg.begin(), wg.end(), [](int a, int c2) { return (a & 63) > (c2 & 63); });
                if (!kv16_on && (int)wg.size() > e->num_cus) std::reverse(wg.begin() + e->num_cus, wg.end());
                for (int v : wg) map[n_live++] = v;
            }
        }
        kv16_on = kv16_on && n_live > 0;
    }
    LCK(hipMemcpyAsync(b.tok, b.h_meta, (size_t)5 * b.cap * 4, hipMemcpyHostToDevice, st));
    {
        BEmbedParams P{e->embd_raw, e->embd_type, D, b.tok, b.x, e->gamma_row, e->gamma_val,
                       b.pos, b.stream, e->rope_cos, e->rope_sin, b.tcos, b.tsin, b.tkv, hd / 2, hd, e->kv_stream_stride};
        hipLaunchKernelGGL(bembed_kernel, dim3(n), dim3(256), 0, st, P);
        LCK(hipGetLastError());
    }
    const int nt16 = ((n + 63) / 64) * 4;
    // position splits any token of this step can see: the attention grids skip the rest (a decode batch at short
    // contexts would otherwise dispatch 15 empty workgroups for every live one)
    const bool no_tile = attn_knobs().no_tile, no_fin = attn_knobs().no_fin;
    int nsplit = 1;
    for (int i = 0; i < n; i++) nsplit = std::max(nsplit, b.h_meta[b.cap + i] / ATT_CH + 1);
    bool consecutive = true;   // positions pos0, pos0 + 1, ...: a prompt
    for (int i = 1; i < n && consecutive; i++) consecutive = b.h_meta[b.cap + i] == b.h_meta[b.cap] + i;
    nsplit = std::min(nsplit, e->nsplit_max);
    const bool tile_attn = one_stream && n >= 8 && attn_tile_supported(e->gqa) && !no_tile;   // prompts: attn_tile16_kernel
    const bool fin_attn = !tile_attn && nsplit == 1 && !no_fin;                                 // one split: attn_kernel<..., FIN>
    const char *rk = getenv("NL_ROPE_IN_ATTN");    // knob (tests, tools; read per step): 0 keeps the brope_kv launch
    const bool rope_attn_knob = !(rk && atoi(rk) == 0);
    GemmOut pend{nullptr, nullptr, 1, 0, nullptr};   // GEMM output not yet folded into the residual stream
    // RMSNorm folded around the GEMMs (QGemmParams::NormOut / NormIn): when WO and down run unsplit on qgemm2_kernel and
    // their consumers are the fused-epilogue GEMMs of the same kernel -- a Q4_0 prompt -- the producing GEMM writes the
    // next GEMM's fragments and per-block sums of squares itself and the two bnorm launches per layer disappear (mini,
    // 2047 tokens: 2 x 8.7 us of 187 us per layer).  Fragment stores then alternate: the consumer of a producing GEMM
    // reads xfrag2, everything else xfrag.
    const char *fk = getenv("NL_FOLD_NORM");   // knob (tests, tools; read per step so a test can flip it): 0 keeps the bnorm launches
    const bool fold_knob = !(fk && atoi(fk) == 0);
    const bool dg = dgemm_step_ok(e, n);      // short token runs (decode batches): nl_dgemm.h, always folded
    // ... and the LM head of a step that wants every token's logits (knob NL_DGEMM_HEAD=0, read per step: the split-K launches)
    const bool dg_head = dg && lm_mode == 1 && e->dg_head && n <= b.lm_cap && (size_t)n * e->lm_head.ntiles * 2 <= b.kpart_cap &&
                         !(getenv("NL_DGEMM_HEAD") && atoi(getenv("NL_DGEMM_HEAD")) == 0);
    bool fold = fold_knob && !c.qk_norm && D % 64 == 0 && c.n_layers > 0;
    for (int l = 0; l < c.n_layers && fold; l++) {
        const nl_engine::Layer &L = e->layers[l];
        fold = qgemm2_plain_unsplit(L.wo.wtype, L.wo.ntiles, n) && qgemm2_plain_unsplit(L.down.wtype, L.down.ntiles, n) &&
               L.wo.ntiles % 4 == 0 && L.down.ntiles % 4 == 0 &&
               qgemm2_ok(L.qkv.wtype, n) && qgemm_rope_fits(L.qkv.ntiles, n) &&
               L.up.wtype == L.gate.wtype && qgemm2_ok(L.gate.wtype, n) && qgemm_swiglu_fits(L.gate.ntiles, n);
    }
    if (dg) fold = true;
    const int nrb = dg ? D / 32 : D / 64;      // partial sums of squares per token
    // NL_PREFILL_PRECISION=fp16x1 (read per step: tests and bench.py flip it): the long-prompt GEMMs (qgemm2_kernel) and the
    // prompt attention multiply only the fp16 hi half of every activation / probability -- half / a third of the matrix work,
    // activations rounded to 11 bits.  The default x = hi + lo keeps float32-grade results (logit tolerance 1e-4).
    const char *pk = getenv("NL_PREFILL_PRECISION");
    const int x1 = pk && std::string(pk) == "fp16x1" ? 1 : 0;
    const int x1_gemm = x1 || (pk && std::string(pk) == "fp16x1-gemm"), x1_attn = x1 || (pk && std::string(pk) == "fp16x1-attn");   // (developer: one half of the mode)
    // (pre-scales: the attention norm's consumer undoes sA and leaves sB for the WO producer; the feed-forward norm's consumer
    //  undoes sB and leaves sA for the down producer; layer 0's unfolded attention norm leaves the first sB)
    float *const sA = b.nscale, *const sB = b.nscale + b.cap;
    const QGemmParams::NormIn nin_attn{b.ssq, nrb, D, c.rms_eps, sA, sB}, nin_ffn{b.ssq, nrb, D, c.rms_eps, sB, sA},
                              nin_off{nullptr, 0, 0, 0.f, nullptr, nullptr};
    auto norm = [&](const float *w, const PackedMat &next, int item0, int cnt, float *scale_out = nullptr) {
        BNormParams P{b.x, pend, w, c.rms_eps, D, item0, b.xfrag, ((cnt + 63) / 64) * 4, next.wtype == WT_Q4_0 ? 1 : 0, scale_out};
        const int nu = D / 8;
        if (nu <= 256) hipLaunchKernelGGL(bnorm_kernel<1>, dim3(cnt), dim3(std::min(256, (nu + 63) / 64 * 64)), 0, st, P);
        else if (nu <= 512) hipLaunchKernelGGL(bnorm_kernel<2>, dim3(cnt), dim3(256), 0, st, P);
        else if (nu <= 1024) hipLaunchKernelGGL(bnorm_kernel<4>, dim3(cnt), dim3(256), 0, st, P);
        else hipLaunchKernelGGL(bnorm_generic_kernel, dim3(cnt), dim3(256), 0, st, P);
        pend = GemmOut{nullptr, nullptr, 1, 0, nullptr};
        return hipGetLastError();
    };
    for (int l = 0; l < c.n_layers; l++) {
        nl_engine::Layer &L = e->layers[l];
        float *kc = e->kcache + (long long)l * e->kv_layer_stride;
        float *vc = e->vcache + (long long)l * e->kv_layer_stride;
        GemmOut qkv_out{nullptr, nullptr, 1, 0, nullptr};
        bool rope_in_attn = false;
        const bool folded_in = fold && l > 0;      // the previous layer's down GEMM wrote this layer's Q|K|V input
        if (!folded_in) LCK(norm(L.attn_norm, L.qkv, 0, n, fold ? sB : nullptr));
        if (dg || (!c.qk_norm && qgemm_rope_fits(L.qkv.ntiles, n))) {
            // Q|K|V, RoPE, biases and the KV store in ONE launch (QK-norm needs whole heads: unfused path)
            QGemmParams P{};
            const PackedMat &m = L.qkv;
            P.q = dg ? m.q3 : m.q; P.s = dg ? m.s3 : m.s; P.rows = m.rows; P.cols = m.cols; P.npairs = m.npairs; P.ntiles = m.ntiles;
            P.xf = folded_in ? b.xfrag2 : b.xfrag; P.nrm_in = folded_in ? nin_attn : nin_off;
            P.n_tokens = n; P.ldo = (int)R; P.x1 = x1_gemm;
            P.rope = QGemmParams::Rope{b.pos, b.stream, e->rope_cos, e->rope_sin, b.q, kc, vc, e->kv_stream_stride,
                                       L.bq, L.bk, L.bv, hd, e->Hs, e->KVs, c.seq_len, c.rope_conjugate, b.tcos, b.tsin, b.tkv};
            LCK(dg ? launch_dgemm_rope(P, st) : launch_qgemm_rope(m.wtype, P, st));
        } else {
            LCK(qg(e, b, L.qkv, n, b.qkv, R, nullptr, st, &qkv_out, b.kpart, nullptr, nullptr, nullptr, x1_gemm));
            // decode batches (every token its own stream: no token of the step attends over another's K / V row), no QK-norm:
            // the attention launch rotates and stores its own q / k / v rows (attn_rope_prologue) -- no brope_kv launch.
            // Below position 128 that is the one-split FIN kernel; with two splits per token every split rotates q and the
            // last one k / v.  Beyond that the repeated q work of every split costs more than the launch it saves (goldie x
            // 64 streams: positions 130..145 2.03 -> 1.92 ms per step, 300.. 2.19 -> 2.19, 1000.. 2.85 -> 2.95).
            rope_in_attn = !tile_attn && nsplit <= 2 && !one_stream && !c.qk_norm && rope_attn_knob;
            if (!rope_in_attn) {
                const GemmOut &qkv = qkv_out;
                BRopeParams P{qkv, R, hd, e->Hs, e->KVs, c.seq_len, c.rope_conjugate, c.qk_norm, c.rms_eps, b.pos, b.stream,
                              e->rope_cos, e->rope_sin, b.q, kc, vc, e->kv_stream_stride, L.bq, L.bk, L.bv};
                // few tokens (decode batches): one element per thread where the row fits, so the per-element chain (slabs ->
                // rotate -> LDS) is paid once (goldie x 64 streams: 15 -> 5.6 us); long prompts keep 256-thread workgroups
                const int threads = n <= 256 ? std::min(1024, std::max(256, (int)((R + 63) / 64 * 64))) : 256;
                hipLaunchKernelGGL(brope_kv_kernel, dim3(n), dim3(threads), (size_t)R * 4, st, P);
                LCK(hipGetLastError());
            }
        }
        {
            AttnParams P{b.q, kc, vc, e->kv_stream_stride, b.part_o, b.part_ml, e->ctl, e->KVs, c.seq_len, e->nsplit_max,
                         (float)(1.0 / std::sqrt((double)hd)), 0, b.pos, b.stream, (long long)HQ,
                         (long long)e->Hs * e->nsplit_max};
            bool merged = false;
            if (one_stream && n >= 8 && attn_tile_supported(e->gqa) && !no_tile) {
                // prefill: the step's tokens share a stream -> K/V split staged once per tile of tokens, fp16 MFMA with hi/lo-split operands (attn_tile16_kernel)
                AttnParams T = P;
                T.x1 = x1_attn;
                T.single_stream = c.max_streams == 1 ? 1 : 0;
                T.pos_base_valid = consecutive ? 1 : 0;   // (a prompt: the kernel derives its key range without reading bpos)
                T.pos_base = b.h_meta[b.cap];
                T.live_map = n_live > 0 ? b.tok + 3 * b.cap : nullptr;
                if (n_live > 0 && kv16_on) {
                    // every chunk the prompt can see, split into fp16 halves once for all the tiles that read it
                    const long long so = T.single_stream ? 0 : (long long)b.h_meta[2 * b.cap] * e->kv_stream_stride;
                    const int n_keys = b.h_meta[b.cap] + n;
                    Kv16BuildParams KB{kc + so, vc + so, b.kv16, c.seq_len, e->nsplit_max, n_keys};
                    const dim3 kg(e->KVs, (n_keys + ATT_CH - 1) / ATT_CH);
                    if (hd == 64) hipLaunchKernelGGL(kv16_build_kernel<64>, kg, dim3(512), 0, st, KB);
                    else hipLaunchKernelGGL(kv16_build_kernel<32>, kg, dim3(512), 0, st, KB);
                    T.kv16 = b.kv16;
                }
                LCK(hd == 64 ? launch_attn_tile_hd<64>(e->gqa, T, n, e->KVs, n_live > 0 ? n_live : nsplit, st)
                             : launch_attn_tile_hd<32>(e->gqa, T, n, e->KVs, n_live > 0 ? n_live : nsplit, st));
            } else if (fin_attn) {
                // every position < 128: one split per row, the attention kernel normalises and writes the fragments
                P.fin_xf = b.xfrag; P.fin_nt16 = nt16; P.fin_q4 = L.wo.wtype == WT_Q4_0 ? 1 : 0;
                if (rope_in_attn)
                    P.rp = AttnParams::Rope{1, qkv_out, (int)R, e->Hs, c.rope_conjugate, e->rope_cos, e->rope_sin, L.bq, L.bk, L.bv, kc, vc};
                LCK(hd == 64 ? launch_attn_fin_hd<64>(e->gqa, P, dim3(e->KVs, 1, n), st)
                             : launch_attn_fin_hd<32>(e->gqa, P, dim3(e->KVs, 1, n), st));
                merged = true;
            } else {
                if (rope_in_attn)
                    P.rp = AttnParams::Rope{1, qkv_out, (int)R, e->Hs, c.rope_conjugate, e->rope_cos, e->rope_sin, L.bq, L.bk, L.bv, kc, vc};
                LCK(launch_attn(hd, e->gqa, P, dim3(e->KVs, nsplit, n), st));
            }
            if (!merged) {
            const bool tile16 = one_stream && n >= 8 && attn_tile_supported(e->gqa) && !no_tile;
            BMergeParams M{b.part_o, b.part_ml, b.pos, tile16 && n_live > 0 ? b.tok + 4 * b.cap : nullptr, e->Hs, e->nsplit_max, hd, b.xfrag, nt16, L.wo.wtype == WT_Q4_0 ? 1 : 0, tile16 ? 1 : 0};
            {
                const long long units = (long long)n * e->Hs * hd / 8;
                if (M.nparts && max_nparts <= 4) hipLaunchKernelGGL(battn_merge_kernel<4>, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, M, n);
                else hipLaunchKernelGGL(battn_merge_kernel<16>, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, M, n);
            }
            }
            LCK(hipGetLastError());
        }
        // dgemm_kernel's producer GEMM: x += W . in (+ bias), the next GEMM's fragments and sums of squares from its epilogue
        auto dg_plain = [&](const PackedMat &m, const uint4 *in, const float *bias, const QGemmParams::NormOut *nout) {
            QGemmParams P{};
            P.q = m.q3; P.s = m.s3; P.rows = m.rows; P.cols = m.cols; P.npairs = m.npairs; P.ntiles = m.ntiles;
            P.xf = in; P.n_tokens = n; P.out = b.x; P.ldo = D; P.resid = b.x; P.bias = bias;
            if (nout) P.nrm_out = *nout;
            pend = GemmOut{b.x, nullptr, 1, (long long)n * D, nullptr};
            return launch_dgemm_plain(P, st);
        };
        if (fold && dg) {
            const QGemmParams::NormOut nout{L.ffn_norm, b.xfrag2, b.ssq, sB};
            LCK(dg_plain(L.wo, b.xfrag, L.bo, &nout));
        } else if (fold) {
            const QGemmParams::NormOut nout{L.ffn_norm, b.xfrag2, b.ssq, sB};
            LCK(qg(e, b, L.wo, n, b.x, D, b.x, st, &pend, b.kpart, L.bo, nullptr, &nout, x1_gemm));
        } else {
            LCK(qg(e, b, L.wo, n, b.x, D, b.x, st, &pend, b.kpart, L.bo, nullptr, nullptr, x1_gemm));
            LCK(norm(L.ffn_norm, L.gate, 0, n));
        }
        if (L.up.wtype != L.gate.wtype)   // (a mixed-type file: the fragment k-slot order differs per type)
            return e->fail(NL_ERR_UNSUPPORTED, "gate and up projections of different quantisation types");
        const uint4 *down_in = b.xfrag;
        if (dg || qgemm_swiglu_fits(L.gate.ntiles, n)) {
            // gate, up and SiLU(gate) * up in ONE launch; h leaves as fragments in the second fragment store
            // (the first is still being read by other workgroups of this launch)
            QGemmParams P{};
            const PackedMat &m = L.gate;
            P.q = dg ? m.q3 : m.q; P.s = dg ? m.s3 : m.s; P.rows = m.rows; P.cols = m.cols; P.npairs = m.npairs; P.ntiles = m.ntiles;
            P.xf = fold ? b.xfrag2 : b.xfrag; P.nrm_in = fold ? nin_ffn : nin_off;
            P.n_tokens = n; P.ldo = e->Is; P.x1 = x1_gemm;
            P.q1 = dg ? L.up.q3 : L.up.q; P.s1 = dg ? L.up.s3 : L.up.s;
            P.xf_out = fold ? b.xfrag : b.xfrag2; P.out_q4 = L.down.wtype == WT_Q4_0 ? 1 : 0;
            LCK(dg ? launch_dgemm_swiglu(P, st) : launch_qgemm_swiglu(m.wtype, P, st));
            down_in = P.xf_out;
        } else {
            GemmOut gate, up;
            {   // gate and up in ONE launch: same input fragments, twice the workgroups, half the split-K
                QGemmParams P{};
                const PackedMat &m = L.gate;
                P.q = m.q; P.s = m.s; P.rows = m.rows; P.cols = m.cols; P.npairs = m.npairs; P.ntiles = m.ntiles;
                P.xf = b.xfrag; P.n_tokens = n; P.out = b.g; P.ldo = e->Is; P.x1 = x1_gemm;
                P.q1 = L.up.q; P.s1 = L.up.s; P.out1 = b.u; P.part1 = b.kpart2;
                int ks = 1;
                LCK(launch_qgemm(m.wtype, P, st, b.kpart, b.kpart_cap, &ks));
                gate = GemmOut{b.g, b.kpart, ks, (long long)n * e->Is, nullptr};
                up = GemmOut{b.u, b.kpart2, ks, (long long)n * e->Is, nullptr};
            }
            BSwigluParams P{gate, up, e->Is, n, b.xfrag, nt16, L.down.wtype == WT_Q4_0 ? 1 : 0};
            const long long tot = (long long)n * e->Is / 8;
            hipLaunchKernelGGL(bswiglu_kernel, dim3((unsigned)std::min<long long>((tot + 255) / 256, 4096)), dim3(256), 0, st, P);
            LCK(hipGetLastError());
        }
        if (dg) {
            // (the last layer: the LM head's input when that runs on dgemm_kernel too -- the final norm's weights)
            const float *const nw = l + 1 < c.n_layers ? e->layers[l + 1].attn_norm : dg_head ? e->output_norm : nullptr;
            const QGemmParams::NormOut nout{nw, b.xfrag2, b.ssq, sA};
            LCK(dg_plain(L.down, down_in, nullptr, nw ? &nout : nullptr));
        } else if (fold && l + 1 < c.n_layers) {
            const nl_engine::Layer &Ln = e->layers[l + 1];
            const QGemmParams::NormOut nout{Ln.attn_norm, b.xfrag2, b.ssq, sA};
            LCK(qg(e, b, L.down, n, b.x, D, b.x, st, &pend, b.kpart, nullptr, down_in, &nout, x1_gemm));
        } else {
            LCK(qg(e, b, L.down, n, b.x, D, b.x, st, &pend, b.kpart, nullptr, down_in, nullptr, x1_gemm));
        }
    }
    if (lm_mode) {
        // logits rows [0, cnt) of bt.logits / ids [0, cnt): all tokens (mode 1, n <= lm_cap) or just the last (mode 2)
        const int first = lm_mode == 2 ? n - 1 : 0, cnt = lm_mode == 2 ? 1 : n;
        if (cnt > b.lm_cap) return e->fail(NL_ERR_INVALID, "LM head batch %d exceeds %d", cnt, b.lm_cap);
        if (dg_head) {
            // final RMSNorm folded (the last down launch wrote the fragments and sums of squares), logits + one argmax candidate
            // per (token, 16 rows) from the GEMM's epilogue, the candidates merged by a launch that reads 1 / 8 of the logits' bytes
            const PackedMat &m = e->lm_head;
            QGemmParams P{};
            P.q = m.q3; P.s = m.s3; P.rows = m.rows; P.cols = m.cols; P.npairs = m.npairs; P.ntiles = m.ntiles;
            P.xf = b.xfrag2; P.nrm_in = nin_attn; P.n_tokens = n; P.out = b.logits; P.ldo = c.vocab; P.part1 = b.kpart;
            LCK(launch_dgemm_head(P, st, e->num_cus));
            hipLaunchKernelGGL(dg_argmax_kernel, dim3(n), dim3(1024), 0, st, reinterpret_cast<const uint2 *>(b.kpart), m.ntiles, b.ids);
        } else {
            LCK(norm(e->output_norm, e->lm_head, first, cnt));
            LCK(launch_qgemm_plain(e, b, e->lm_head, cnt, b.logits, c.vocab, st));
            hipLaunchKernelGGL(bargmax_kernel, dim3(cnt), dim3(1024), 0, st, b.logits, c.vocab, b.ids);
        }
        LCK(hipGetLastError());
    }
#undef LCK
    return NL_OK;
}

}  // namespace

// ---- one-process tensor-parallel group (nl_create_group) ---------------------------------------------------------------------
struct GroupCtl {
    std::vector<nl_engine *> members;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::function<int(nl_engine *, int)> job;
    unsigned long long generation = 0;
    int pending = 0;
    bool quit = false;
    std::vector<int> rc;
};

namespace {

void group_worker(GroupCtl *g, int r) {
    unsigned long long seen = 0;
    for (;;) {
        std::function<int(nl_engine *, int)> job;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_job.wait(lk, [&] { return g->quit || g->generation != seen; });
            if (g->quit) return;
            seen = g->generation;
            job = g->job;
        }
        const int rc = job(g->members[r], r);
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->rc[r] = rc;
            if (--g->pending == 0) g->cv_done.notify_all();
        }
    }
}

// run f(member, rank) on every rank's thread and wait for all; the first failing rank's code and message become the leader's
int group_run(nl_engine *lead, std::function<int(nl_engine *, int)> f) {
    GroupCtl *g = lead->grp;
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->job = std::move(f);
        g->pending = (int)g->members.size();
        g->generation++;
        g->cv_job.notify_all();
        g->cv_done.wait(lk, [&] { return g->pending == 0; });
    }
    for (size_t r = 0; r < g->members.size(); r++)
        if (g->rc[r] != NL_OK) {
            lead->err = "rank " + std::to_string(r) + ": " + g->members[r]->err;
            return g->rc[r];
        }
    // a warning a rank left while returning NL_OK (a retired fused plan) is the leader's note too
    lead->err = g->members[0]->err;
    return NL_OK;
}

void group_stop(GroupCtl *g) {
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->quit = true;
        g->cv_job.notify_all();
    }
    for (auto &t : g->workers) if (t.joinable()) t.join();
}

int p2p_alloc_area(nl_engine *e);   // below (shared with nl_p2p_export)

}  // namespace

// ============================================================== C ABI =====

extern "C" {

int nl_abi_version(void) { return 1; }

#ifndef NL_SRC_SHA
#define NL_SRC_SHA "unknown"
#endif
#ifndef NL_GIT_HEAD
#define NL_GIT_HEAD "unknown"
#endif
const char *nl_build_info(void) { return "src=" NL_SRC_SHA " git=" NL_GIT_HEAD; }

#ifdef NL_TP_STAMPS
__attribute__((visibility("default"))) int nl_debug_tp_stamps(long long *out) {   // developer build only: 8 x 16 values
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nl::g_tp_stamps), 8 * 16 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
__attribute__((visibility("default"))) int nl_debug_tp_census(long long *out) {   // 2 x 512 x 2 values
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nl::g_tp_census), 2 * 512 * 2 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef NL_ATTN_STAMPS
__attribute__((visibility("default"))) int nl_debug_attn_stamps(long long *out) {   // developer build only: 16 values
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nl::g_attn_stamps), 16 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef NL_ATT_STAMPS
// developer build only (tools/att_stamps.sh): phase stamps of one workgroup of the prompt attention kernel
__attribute__((visibility("default"))) int nl_debug_att_stamps(long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nl::g_att_stamps), 64 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
__attribute__((visibility("default"))) int nl_debug_att_timeline(long long *out) {   // 512 x 64 values
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nl::g_att_timeline), 512 * 64 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
__attribute__((visibility("default"))) int nl_debug_att_census(long long *out) {   // 4 x 8192 values
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nl::g_att_census), 4 * 8192 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef DG_STAMPS
__attribute__((visibility("default"))) int nl_debug_dg_stamps(long long *stamps64, long long *census4096) {
    if (hipMemcpyFromSymbol(stamps64, HIP_SYMBOL(nl::g_dg_stamps), 64 * sizeof(long long)) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(census4096, HIP_SYMBOL(nl::g_dg_census), 2 * 2048 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
__attribute__((visibility("default"))) int nl_debug_dg_log(long long *log16384, unsigned *n) {
    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(nl::g_dg_log_n), sizeof(unsigned)) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(log16384, HIP_SYMBOL(nl::g_dg_log), 4 * 4096 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

int nl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *nl_last_error(nl_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int nl_create(const nl_config *cfg, nl_handle *out) {
    if (!cfg || !out) { g_create_error = "null argument"; return NL_ERR_INVALID; }
    *out = nullptr;
    nl_config c = *cfg;
    if (c.head_dim == 0 && c.n_heads > 0) c.head_dim = c.dim / c.n_heads;  // go/model.go:140-142
    if (c.seq_len > 2048) c.seq_len = 2048;                                // go/model.go:145-148
    if (c.max_streams <= 0) c.max_streams = 1;
    if (c.tp_size <= 0) c.tp_size = 1;
    auto bad = [&](const char *m) { g_create_error = m; return NL_ERR_INVALID; };
    if (c.n_layers <= 0 || c.dim <= 0 || c.n_heads <= 0 || c.n_kv_heads <= 0 || c.interm <= 0 || c.vocab <= 0 ||
        c.seq_len <= 0)
        return bad("config: non-positive dimension");
    if (c.n_heads % c.n_kv_heads) return bad("config: n_heads not a multiple of n_kv_heads");
    if (c.dim % 32 || c.interm % 32) return bad("config: dim and interm must be multiples of 32 (whole quant blocks)");
    if (c.head_dim != 32 && c.head_dim != 64) { g_create_error = "head_dim must be 32 or 64 on this build"; return NL_ERR_UNSUPPORTED; }
    int gq = c.n_heads / c.n_kv_heads;
    if (gq != 1 && gq != 2 && gq != 3 && gq != 4 && gq != 8) { g_create_error = "GQA group size must be 1,2,3,4 or 8"; return NL_ERR_UNSUPPORTED; }
    if (c.tp_rank < 0 || c.tp_rank >= c.tp_size) return bad("config: tp_rank out of range");
    if (c.n_heads % c.tp_size || c.n_kv_heads % c.tp_size || c.vocab % c.tp_size || c.interm % (32 * c.tp_size))
        return bad("config: heads / kv heads / vocab / interm(32-blocks) must divide by tp_size");
    int ndev = 0;
    hipError_t s = hipGetDeviceCount(&ndev);
    if (s != hipSuccess || ndev == 0) {
        g_create_error = std::string("no HIP device: ") + hipGetErrorString(s);
        return NL_ERR_HIP;
    }
    if (c.device < 0 || c.device >= ndev) return bad("config: device ordinal out of range");
    nl_engine *e = new nl_engine();
    e->cfg = c;
    e->dev = c.device;
    e->G = c.tp_size; e->rank = c.tp_rank;
    e->hd = c.head_dim; e->gqa = gq;
    e->Hs = c.n_heads / e->G; e->KVs = c.n_kv_heads / e->G; e->Is = c.interm / e->G; e->Vs = c.vocab / e->G;
    e->nsplit_max = (c.seq_len + ATT_CH - 1) / ATT_CH;
    e->layers.resize(c.n_layers);
    e->use_graph = !(c.flags & NL_FLAG_NO_GRAPH) && !getenv("NL_NO_GRAPH");
    if (getenv("NL_FORCE_TP_PLAN")) e->force_tp_plan = true;
    {
        const char *pk = getenv("NL_PERSIST");      // knob (tests, tools): 0 keeps the launch plans for greedy chains too
        e->pd.candidate = !(pk && atoi(pk) == 0) && c.tp_size == 1 && !(c.flags & NL_FLAG_LOCAL_GROUP) && !c.qk_norm &&
                          pd_shape_ok(c.dim, c.interm, c.n_heads, c.n_kv_heads, c.head_dim, c.vocab, c.n_layers);
    }
    if (const char *v = getenv("NL_SUB_BATCHES")) e->sub_batches = std::max(1, std::min(4, atoi(v)));   // knob (tests, tools)
    if (const char *v = getenv("NL_TW")) e->tw_override = atoi(v);
    if (const char *v = getenv("NL_KW")) e->kw_override = atoi(v);
    if ((s = hipSetDevice(e->dev)) != hipSuccess || (s = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess ||
        (s = hipEventCreate(&e->ev0)) != hipSuccess || (s = hipEventCreate(&e->ev1)) != hipSuccess) {
        g_create_error = std::string("device init: ") + hipGetErrorString(s);
        delete e;
        return NL_ERR_HIP;
    }
    *out = e;
    return NL_OK;
}

// LoadLlamaModel for a model sharded over the GPUs of ONE process (go/main.go:63 is one process; BASELINE's 7.9B tier shards
// across the node's 8 GPUs): the returned handle is used exactly like nl_create's -- nl_upload_tensor takes the FULL tensors
// (every rank keeps its slice), nl_finalize, nl_forward, nl_decode_greedy, nl_sample_decode, nl_prefill, nl_reset,
// nl_destroy ... -- and steps n rank engines (tp_rank r on device_ids[r]) behind it: row / column tensor parallelism with
// the push all-reduce of nl_p2p.h between them, the receive areas reached through plain peer pointers
// (hipDeviceEnablePeerAccess) instead of hipIpc handles.  device_ids may repeat (all ranks on one device: the one-GPU test
// configuration, bitwise equal to the in-process shard group).
int nl_create_group(const nl_config *cfg, const int *device_ids, int n, nl_handle *out) {
    if (!cfg || !device_ids || !out) { g_create_error = "null argument"; return NL_ERR_INVALID; }
    *out = nullptr;
    if (n != 2 && n != 4 && n != 8) { g_create_error = "nl_create_group: 2, 4 or 8 ranks"; return NL_ERR_UNSUPPORTED; }
    if (cfg->tp_size > 1 && cfg->tp_size != n) { g_create_error = "nl_create_group: tp_size disagrees with the device list"; return NL_ERR_INVALID; }
    nl_engine *lead = new nl_engine();
    GroupCtl *g = new GroupCtl();
    lead->grp = g;
    auto fail = [&](int rc, const std::string &msg) {
        for (nl_engine *m : g->members) nl_destroy(m);
        delete g;
        delete lead;
        g_create_error = msg;
        return rc;
    };
    for (int r = 0; r < n; r++) {
        nl_config c = *cfg;
        c.tp_size = n; c.tp_rank = r; c.device = device_ids[r];
        c.flags &= ~(NL_FLAG_LOCAL_GROUP | NL_FLAG_GROUP_FUSED);
        nl_engine *m = nullptr;
        const int rc = nl_create(&c, &m);
        if (rc != NL_OK) return fail(rc, "rank " + std::to_string(r) + ": " + g_create_error);
        g->members.push_back(m);
    }
    lead->cfg = g->members[0]->cfg;          // the effective configuration (seq_len cap, head_dim) as the caller sees it:
    lead->cfg.tp_size = 1; lead->cfg.tp_rank = 0; lead->cfg.device = device_ids[0];   // ... one model
    lead->dev = device_ids[0];
    // peers: every rank's device may store into every other rank's receive area
    for (int a = 0; a < n; a++)
        for (int b = 0; b < n; b++) {
            if (device_ids[a] == device_ids[b]) continue;
            int can = 0;
            if (hipSetDevice(device_ids[a]) != hipSuccess || hipDeviceCanAccessPeer(&can, device_ids[a], device_ids[b]) != hipSuccess || !can)
                return fail(NL_ERR_UNSUPPORTED, "device " + std::to_string(device_ids[a]) + " cannot reach device " + std::to_string(device_ids[b]) + " (no peer access)");
            const hipError_t s = hipDeviceEnablePeerAccess(device_ids[b], 0);
            if (s != hipSuccess && s != hipErrorPeerAccessAlreadyEnabled) return fail(NL_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(s));
            (void)hipGetLastError();
        }
    for (int r = 0; r < n; r++)
        if (int rc = p2p_alloc_area(g->members[r])) return fail(rc, "rank " + std::to_string(r) + ": " + g->members[r]->err);
    for (int r = 0; r < n; r++) {
        nl_engine::P2P &p = g->members[r]->p2p;
        for (int q = 0; q < n; q++) p.peer[q] = g->members[q]->p2p.area;       // plain peer pointers: one address space in one process
        if (!p.uncached)      // NL_P2P_CACHED test knob: only sound while every rank shares this device's L2 (as nl_p2p_import checks)
            for (int q = 0; q < n; q++)
                if (device_ids[q] != device_ids[r]) return fail(NL_ERR_UNSUPPORTED, "NL_P2P_CACHED is a one-device test knob: rank " + std::to_string(q) + " lives on device " + std::to_string(device_ids[q]));
        p.on = true;
    }
    g->rc.assign(n, NL_OK);
    for (int r = 0; r < n; r++) g->workers.emplace_back(group_worker, g, r);
    *out = lead;
    return NL_OK;
}

int nl_get_config(nl_handle h, nl_config *out) {
    if (!h || !out) return NL_ERR_INVALID;
    *out = h->cfg;
    return NL_OK;
}

int nl_upload_tensor(nl_handle e, const char *name, uint32_t type, const void *data, uint64_t nbytes, uint64_t rows,
                     uint64_t cols) {
    if (e && e->grp) return group_run(e, [&](nl_engine *m, int) { return nl_upload_tensor(m, name, type, data, nbytes, rows, cols); });
    if (!e || !name || !data) return NL_ERR_INVALID;
    if (e->finalized) return e->fail(NL_ERR_STATE, "upload after nl_finalize");
    const nl_config &c = e->cfg;
    Slot sl;
    parse_name(name, sl);
    if (sl.layer >= c.n_layers) return e->fail(NL_ERR_INVALID, "tensor %s: layer out of range", name);
    if (!type_supported(type)) return e->fail(NL_ERR_UNSUPPORTED, "tensor %s: ggml type %u is not supported on device", name, type);
    if (raw_bytes(type, rows * cols) != nbytes)
        return e->fail(NL_ERR_INVALID, "tensor %s: %llu bytes given, %zu expected for %llux%llu type %u", name,
                       (unsigned long long)nbytes, raw_bytes(type, rows * cols), (unsigned long long)rows,
                       (unsigned long long)cols, type);
    HIPCK(e, hipSetDevice(e->dev));
    const std::string &f = sl.field;

    auto norm_upload = [&](float **dst) -> int {
        if ((int)(rows * cols) != c.dim) return e->fail(NL_ERR_INVALID, "tensor %s: expected %d elements", name, c.dim);
        std::vector<float> tmp(c.dim);
        if (type == WT_F32) memcpy(tmp.data(), data, (size_t)c.dim * 4);
        else if (type == WT_F16) {
            const uint16_t *hp = (const uint16_t *)data;
            for (int i = 0; i < c.dim; i++) tmp[i] = __half2float(__ushort_as_half(hp[i]));
        } else return e->fail(NL_ERR_UNSUPPORTED, "tensor %s: norm weights must be F32 or F16", name);
        if (!*dst) HIPCK(e, dalloc(dst, (size_t)c.dim, &e->bytes_weights));
        HIPCK(e, hipMemcpy(*dst, tmp.data(), (size_t)c.dim * 4, hipMemcpyHostToDevice));
        return NL_OK;
    };
    // matrix slot: copy raw bytes to the device, re-pack this rank's slice
    auto matrix_upload = [&](PackedMat &m, int exp_rows, int exp_cols, int row0, int nrows, int col0, int ncols,
                             int tile0, int tiles_total, int rowmap, bool first) -> int {
        if ((int)rows != exp_rows || (int)cols != exp_cols)
            return e->fail(NL_ERR_INVALID, "tensor %s: shape %llux%llu, expected %dx%d", name, (unsigned long long)rows,
                           (unsigned long long)cols, exp_rows, exp_cols);
        if (first) {
            HIPCK(e, alloc_packed(e, m, (int)type, tiles_total, rowmap == ROWMAP_HEADPERM ? tiles_total * TR : nrows, ncols));
        } else if (m.src_type != (int)type) {
            return e->fail(NL_ERR_UNSUPPORTED, "tensor %s: q/k/v (or gate/up) of one layer must share a type", name);
        }
        if (is_kquant(type) && (exp_cols % 256 || col0 % 256 || ncols % 256))
            return e->fail(NL_ERR_INVALID, "tensor %s: K-quant rows must be whole 256-column super blocks (cols %d, slice %d+%d)",
                           name, exp_cols, col0, ncols);
        // row-sliced tensors (tensor-parallel shards) only move their own rows to the device
        const size_t row_bytes = raw_bytes(type, (uint64_t)exp_cols);
        const uint8_t *src = (const uint8_t *)data + (size_t)row0 * row_bytes;
        const size_t copy_bytes = (size_t)nrows * row_bytes;
        HIPCK(e, stage_reserve(e, copy_bytes));
        hipError_t s = hipMemcpyAsync(e->stage, src, copy_bytes, hipMemcpyHostToDevice, e->stream);
        if (s == hipSuccess) s = repack(e, m, e->stage, (int)type, exp_cols, 0, nrows, col0, ncols, tile0,
                                        rowmap == ROWMAP_HEADPERM ? nrows / TR : (nrows + TR - 1) / TR, rowmap);
        if (s == hipSuccess) s = hipStreamSynchronize(e->stream);
        if (s != hipSuccess) return e->fail(NL_ERR_HIP, "upload %s: %s", name, hipGetErrorString(s));
        return NL_OK;
    };

    if (e->pd.candidate) {
        // the persistent decode (nl_persist.h) packs its own lane images from the raw Q8_0 / Q4_0 / Q5_0 tensors at nl_finalize: keep them
        static const char *const kMat[7] = {"attn_q.weight", "attn_k.weight", "attn_v.weight", "attn_output.weight", "ffn_gate.weight",
                                            "ffn_up.weight", "ffn_down.weight"};
        int slot = -1;
        for (int i = 0; i < 7; i++) if (sl.layer >= 0 && f == kMat[i]) slot = i;
        const bool is_lm = sl.layer < 0 && f == "output.weight";
        const bool is_bias = f.size() > 5 && f.compare(f.size() - 5, 5, ".bias") == 0;
        const bool int8_blocks = type == WT_Q8_0 || type == WT_Q4_0 || type == WT_Q5_0;     // 32 int8-valued quants x an fp16 d
        if (((slot >= 0 || is_lm) && !int8_blocks) || is_bias) {
            e->pd.candidate = false;
        } else if (slot >= 0 || is_lm) {
            uint8_t **dst = is_lm ? &e->pd.lm_raw : &e->pd.raw[sl.layer][slot];
            (is_lm ? e->pd.lm_type : e->pd.rtype[sl.layer][slot]) = (unsigned char)type;
            if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
            HIPCK(e, hipMalloc((void **)dst, nbytes));
            HIPCK(e, hipMemcpy(*dst, data, nbytes, hipMemcpyHostToDevice));
        }
    }
    const int hd = e->hd, D = c.dim;
    if (sl.layer < 0) {
        if (f == "token_embd.weight") {
            if ((int)rows != c.vocab || (int)cols != D) return e->fail(NL_ERR_INVALID, "token_embd.weight: bad shape");
            if (is_kquant(type) && D % 256) return e->fail(NL_ERR_INVALID, "token_embd.weight: K-quant rows need dim %% 256 == 0");
            if (e->embd_raw) hipFree(e->embd_raw);
            HIPCK(e, hipMalloc((void **)&e->embd_raw, nbytes));
            HIPCK(e, hipMemcpy(e->embd_raw, data, nbytes, hipMemcpyHostToDevice));
            e->embd_type = (int)type; e->embd_bytes = nbytes; e->bytes_weights += nbytes;
            return NL_OK;
        }
        if (f == "output_norm.weight") return norm_upload(&e->output_norm);
        if (f == "output.weight") {
            int rc = matrix_upload(e->lm_head, c.vocab, D, e->rank * e->Vs, e->Vs, 0, D, 0, (e->Vs + TR - 1) / TR,
                                   ROWMAP_IDENT, true);
            if (rc) return rc;
            e->lm_head.ready = true; e->have_output = true;
            return NL_OK;
        }
        return e->fail(NL_ERR_INVALID, "unknown tensor %s", name);
    }
    nl_engine::Layer &L = e->layers[sl.layer];
    if (f == "attn_norm.weight") return norm_upload(&L.attn_norm);
    if (f == "ffn_norm.weight") return norm_upload(&L.ffn_norm);
    const int tph = hd / TR;  // tiles per head
    const int qkv_tiles = (e->Hs + 2 * e->KVs) * tph;
    const bool qkv_first = !L.have_q && !L.have_k && !L.have_v;
    if (f == "attn_q.weight") {
        int rc = matrix_upload(L.qkv, c.n_heads * hd, D, e->rank * e->Hs * hd, e->Hs * hd, 0, D, 0, qkv_tiles,
                               ROWMAP_HEADPERM, qkv_first);
        if (!rc) L.have_q = true;
        return rc;
    }
    if (f == "attn_k.weight") {
        int rc = matrix_upload(L.qkv, c.n_kv_heads * hd, D, e->rank * e->KVs * hd, e->KVs * hd, 0, D, e->Hs * tph,
                               qkv_tiles, ROWMAP_HEADPERM, qkv_first);
        if (!rc) L.have_k = true;
        return rc;
    }
    if (f == "attn_v.weight") {
        int rc = matrix_upload(L.qkv, c.n_kv_heads * hd, D, e->rank * e->KVs * hd, e->KVs * hd, 0, D,
                               (e->Hs + e->KVs) * tph, qkv_tiles, ROWMAP_HEADPERM, qkv_first);
        if (!rc) L.have_v = true;
        return rc;
    }
    if (f == "attn_output.weight") {
        int rc = matrix_upload(L.wo, D, c.n_heads * hd, 0, D, e->rank * e->Hs * hd, e->Hs * hd, 0, (D + TR - 1) / TR,
                               ROWMAP_IDENT, true);
        if (!rc) L.wo.ready = true;
        // small models also keep WO as per-head 64-column slices for the fused attention block (nl_block.h): head h =
        // D/16 tiles of one pair each.  The raw tensor is still in the staging buffer.
        const int dt = device_type((int)type);
        if (!rc && e->G == 1 && hd == 64 && D % PAIR == 0 && D <= BLK_MAXG * KL * PAIR && e->Hs <= BLK_MAX_PARTS &&
            (dt == WT_Q8_0 || dt == WT_Q4_0) && !L.wo_head.ready) {
            PackedMat &m = L.wo_head;
            HIPCK(e, alloc_packed(e, m, (int)type, e->Hs * (D / TR), D, hd));
            for (int hh = 0; hh < e->Hs; hh++)
                HIPCK(e, repack(e, m, e->stage, (int)type, c.n_heads * hd, 0, D, hh * hd, hd, hh * (D / TR), D / TR, ROWMAP_IDENT));
            HIPCK(e, hipStreamSynchronize(e->stream));
            m.ready = true;
        }
        return rc;
    }
    if (f == "ffn_gate.weight" || f == "ffn_up.weight") {
        PackedMat &m = f == "ffn_gate.weight" ? L.gate : L.up;
        int rc = matrix_upload(m, c.interm, D, e->rank * e->Is, e->Is, 0, D, 0, (e->Is + TR - 1) / TR, ROWMAP_IDENT, true);
        if (!rc) m.ready = true;
        return rc;
    }
    if (f == "ffn_down.weight") {
        int rc = matrix_upload(L.down, D, c.interm, 0, D, e->rank * e->Is, e->Is, 0, (D + TR - 1) / TR, ROWMAP_IDENT, true);
        if (!rc) L.down.ready = true;
        // small models also keep W_down sliced by 256 columns for the fused feed-forward block (nl_block.h): slice c =
        // D/16 tiles of four pairs each.  The raw tensor is still in the staging buffer.
        const int dt = device_type((int)type);
        if (!rc && e->G == 1 && hd == 64 && D % PAIR == 0 && D <= BLK_MAXG * KL * PAIR && D % (FFN_MEMBERS * 2) == 0 && e->Hs <= BLK_MAX_PARTS &&
            c.interm % FFN_SLICE == 0 && c.interm / FFN_SLICE <= FFN_MAX_PARTS && (dt == WT_Q8_0 || dt == WT_Q4_0) && !L.dn_slice.ready) {
            PackedMat &m = L.dn_slice;
            const int ns = c.interm / FFN_SLICE;
            HIPCK(e, alloc_packed(e, m, (int)type, ns * (D / TR), D, FFN_SLICE));
            for (int sc = 0; sc < ns; sc++)
                HIPCK(e, repack(e, m, e->stage, (int)type, c.interm, 0, D, sc * FFN_SLICE, FFN_SLICE, sc * (D / TR), D / TR, ROWMAP_IDENT));
            HIPCK(e, hipStreamSynchronize(e->stream));
            m.ready = true;
        }
        return rc;
    }
    if (f == "attn_q.bias" || f == "attn_k.bias" || f == "attn_v.bias" || f == "attn_output.bias") {
        // getF32TensorOptional go/model.go:244-247; only this rank's heads are kept, the output bias lives on rank 0
        const bool isq = f == "attn_q.bias", iso = f == "attn_output.bias";
        const int full = isq ? c.n_heads * hd : iso ? D : c.n_kv_heads * hd;
        const int loc = isq ? e->Hs * hd : iso ? D : e->KVs * hd;
        const int off = iso ? 0 : e->rank * loc;
        if ((int)(rows * cols) != full) return e->fail(NL_ERR_INVALID, "tensor %s: expected %d elements", name, full);
        std::vector<float> tmp(full);
        if (type == WT_F32) memcpy(tmp.data(), data, (size_t)full * 4);
        else if (type == WT_F16) {
            const uint16_t *hp = (const uint16_t *)data;
            for (int i = 0; i < full; i++) tmp[i] = __half2float(__ushort_as_half(hp[i]));
        } else return e->fail(NL_ERR_UNSUPPORTED, "tensor %s: biases must be F32 or F16", name);
        if (iso && e->rank != 0) std::fill(tmp.begin(), tmp.end(), 0.f);
        float **dst = isq ? &L.bq : f == "attn_k.bias" ? &L.bk : f == "attn_v.bias" ? &L.bv : &L.bo;
        if (!*dst) HIPCK(e, dalloc(dst, (size_t)loc, &e->bytes_weights));
        HIPCK(e, hipMemcpy(*dst, tmp.data() + off, (size_t)loc * 4, hipMemcpyHostToDevice));
        return NL_OK;
    }
    return e->fail(NL_ERR_INVALID, "unknown tensor %s", name);
}

namespace {
// modes 3 / 4: how many blocks of the projection + attention + WO launch stay until their rows are done (live: they hold projection
// tiles of a kv group that exists; WO owners: b * tpw < WO tiles) -- all of them must fit the compute units at once
bool stayers_fit(const nl_engine *e, int members, int grid, int tpw, int wo_ntiles) {
    int stay = 0;
    for (int b = 0; b < grid; b++) {
        const int cluster = (b / (8 * members)) * 8 + (b & 7);
        const bool live = b < grp_grid(e->KVs, members) && cluster < e->KVs;
        if (live || (long long)b * tpw < wo_ntiles) stay++;
    }
    return stay <= e->num_cus;
}
}  // namespace

int nl_finalize(nl_handle e) {
    if (e && e->grp) {
        if (e->finalized) return e->fail(NL_ERR_STATE, "nl_finalize called twice");
        // one rank after the other (allocations and graph captures; nothing here waits for a peer)
        for (size_t r = 0; r < e->grp->members.size(); r++) {
            const int rc = nl_finalize(e->grp->members[r]);
            if (rc != NL_OK) { e->err = "rank " + std::to_string(r) + ": " + e->grp->members[r]->err; return rc; }
        }
        e->finalized = true;
        return NL_OK;
    }
    if (!e) return NL_ERR_INVALID;
    if (e->finalized) return e->fail(NL_ERR_STATE, "nl_finalize called twice");
    const nl_config &c = e->cfg;
    HIPCK(e, hipSetDevice(e->dev));
    if (!e->embd_raw) return e->fail(NL_ERR_MISSING, "token_embd.weight: tensor not found");
    if (!e->output_norm) return e->fail(NL_ERR_MISSING, "output_norm.weight: tensor not found");
    for (int l = 0; l < c.n_layers; l++) {
        nl_engine::Layer &L = e->layers[l];
        const char *miss = !L.attn_norm ? "attn_norm" : !L.ffn_norm ? "ffn_norm" : !L.have_q ? "attn_q" : !L.have_k ? "attn_k"
                         : !L.have_v ? "attn_v" : !L.wo.ready ? "attn_output" : !L.gate.ready ? "ffn_gate"
                         : !L.up.ready ? "ffn_up" : !L.down.ready ? "ffn_down" : nullptr;
        if (miss) return e->fail(NL_ERR_MISSING, "layer %d %s: tensor not found", l, miss);
        if (L.bq || L.bk || L.bv) {
            // each bias is independently optional in the reference (getF32TensorOptional, go/model.go:244-247):
            // an absent one is a zero vector here
            struct { float **p; int n; } want[3] = {{&L.bq, e->Hs * e->hd}, {&L.bk, e->KVs * e->hd}, {&L.bv, e->KVs * e->hd}};
            for (auto &w : want)
                if (!*w.p) {
                    HIPCK(e, dalloc(w.p, (size_t)w.n, &e->bytes_weights));
                    HIPCK(e, hipMemset(*w.p, 0, (size_t)w.n * 4));
                }
        }
        if (L.gate.src_type != L.up.src_type) return e->fail(NL_ERR_UNSUPPORTED, "layer %d: ffn_gate and ffn_up types differ", l);
    }
    if (!e->have_output) {
        // tied embeddings: output.weight missing -> LM head reads token_embd (go/model.go:195-201)
        PackedMat &m = e->lm_head;
        HIPCK(e, alloc_packed(e, m, e->embd_type, (e->Vs + TR - 1) / TR, e->Vs, c.dim));
        HIPCK(e, repack(e, m, e->embd_raw, e->embd_type, c.dim, e->rank * e->Vs, e->Vs, 0, c.dim, 0, m.ntiles, ROWMAP_IDENT));
        HIPCK(e, hipStreamSynchronize(e->stream));
        m.ready = true;
    }
    // precomputeRoPE go/model.go:346-358 (float64 math, cast to float32)
    const int half = e->hd / 2;
    std::vector<float> hc((size_t)c.seq_len * half), hs((size_t)c.seq_len * half);
    const double theta = (double)c.rope_theta;
    for (int pos = 0; pos < c.seq_len; pos++)
        for (int i = 0; i < half; i++) {
            double freq = 1.0 / std::pow(theta, (double)(2 * i) / (double)e->hd);
            double angle = (double)pos * freq;
            hc[(size_t)pos * half + i] = (float)std::cos(angle);
            hs[(size_t)pos * half + i] = (float)std::sin(angle);
        }
    HIPCK(e, dalloc(&e->rope_cos, hc.size(), &e->bytes_state));
    HIPCK(e, dalloc(&e->rope_sin, hs.size(), &e->bytes_state));
    HIPCK(e, hipMemcpy(e->rope_cos, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    HIPCK(e, hipMemcpy(e->rope_sin, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    // allocState go/model.go:324-343
    HIPCK(e, dalloc(&e->x[0], (size_t)c.dim, &e->bytes_state));
    HIPCK(e, dalloc(&e->x[1], (size_t)c.dim, &e->bytes_state));
    HIPCK(e, dalloc(&e->qbuf, (size_t)e->Hs * e->hd, &e->bytes_state));
    HIPCK(e, dalloc(&e->part_o, (size_t)e->Hs * e->nsplit_max * e->hd, &e->bytes_state));
    HIPCK(e, dalloc(&e->part_ml, (size_t)e->Hs * e->nsplit_max * 2, &e->bytes_state));
    HIPCK(e, dalloc(&e->hb, (size_t)e->Is, &e->bytes_state));
    HIPCK(e, dalloc(&e->ar, (size_t)c.dim, &e->bytes_state));
    if (e->p2p.on) e->logits = reinterpret_cast<float *>((char *)e->p2p.area + e->p2p.off_logits);   // gathered by the peers' LM heads
    else HIPCK(e, dalloc(&e->logits, (size_t)c.vocab, &e->bytes_state));
    e->kv_layer_stride = (long long)e->KVs * c.seq_len * e->hd;
    e->kv_stream_stride = e->kv_layer_stride * c.n_layers;
    const size_t kvn = (size_t)e->kv_stream_stride * c.max_streams;
    HIPCK(e, dalloc(&e->kcache, kvn, &e->bytes_kv));
    HIPCK(e, dalloc(&e->vcache, kvn, &e->bytes_kv));
    HIPCK(e, hipMemsetAsync(e->kcache, 0, kvn * 4, e->stream));
    HIPCK(e, hipMemsetAsync(e->vcache, 0, kvn * 4, e->stream));
    e->hw.assign(c.max_streams, 0);
    e->ids_cap = c.seq_len + 1;
    HIPCK(e, dalloc(&e->ctl, (size_t)CTL_WORDS, &e->bytes_state));
    HIPCK(e, dalloc(&e->ids, (size_t)e->ids_cap, &e->bytes_state));
    HIPCK(e, dalloc(&e->result, (size_t)1, &e->bytes_state));
    e->amax_slots = ((e->Vs + TR - 1) / TR) + 64;  // >= one slot per wave of LM-head rows for any tw
    HIPCK(e, dalloc(&e->amax_val, (size_t)e->amax_slots, &e->bytes_state));
    HIPCK(e, dalloc(&e->amax_idx, (size_t)e->amax_slots, &e->bytes_state));
    HIPCK(e, hipHostMalloc((void **)&e->h_ctl, CTL_WORDS * sizeof(int), hipHostMallocDefault));
    memset(e->h_ctl, 0, CTL_WORDS * sizeof(int));
    if (hipHostMalloc((void **)&e->h_box, 16 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
        hipHostMalloc((void **)&e->h_done, 2 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
        hipHostGetDevicePointer((void **)&e->d_box, e->h_box, 0) == hipSuccess && hipHostGetDevicePointer((void **)&e->d_done, e->h_done, 0) == hipSuccess) {
        memset(e->h_box, 0, 16 * sizeof(int));
        e->h_done[0] = e->h_done[1] = 0;
    } else { e->d_box = nullptr; e->d_done = nullptr; (void)hipGetLastError(); }
    if (e->h_logits) { hipHostFree(e->h_logits); e->h_logits = nullptr; }
    HIPCK(e, hipHostMalloc((void **)&e->h_logits, ((size_t)c.vocab + 4) * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));   // (the resident session's LM-head units store logits straight into it)
    e->d_h_logits = nullptr;
    if (hipHostGetDevicePointer((void **)&e->d_h_logits, e->h_logits, 0) != hipSuccess) { (void)hipGetLastError(); e->d_h_logits = nullptr; }
    if (getenv("NL_NO_HOST_LOGITS")) e->d_h_logits = nullptr;      // knob (A/B): the DMA read-back
    e->ctl_ring_cap = std::max(c.seq_len, c.max_streams);
    HIPCK(e, hipHostMalloc((void **)&e->h_ctl_ring, (size_t)e->ctl_ring_cap * CTL_WORDS * sizeof(int), hipHostMallocDefault));
    memset(e->h_ctl_ring, 0, (size_t)e->ctl_ring_cap * CTL_WORDS * sizeof(int));       // (the callers fill five of the eight words)
    HIPCK(e, hipMemcpy(e->ctl, e->h_ctl, CTL_WORDS * sizeof(int), hipMemcpyHostToDevice));
    if (e->G > 1 && !e->comm && !e->p2p.on && !(c.flags & NL_FLAG_LOCAL_GROUP))
        return e->fail(NL_ERR_STATE, "tp_size %d needs nl_comm_init or nl_p2p_import before nl_finalize", e->G);
    if (e->stage) { hipFree(e->stage); e->stage = nullptr; e->stage_cap = 0; }
    {
        // fused attention block (nl_block.h): every layer must have its per-head WO slices, and the models it pays
        // for are the ones whose per-head weights are small (see the header of nl_block.h)
        {
            hipDeviceProp_t prop{};
            if (hipGetDeviceProperties(&prop, e->dev) == hipSuccess && prop.multiProcessorCount > 0) e->num_cus = prop.multiProcessorCount;
            if (const char *sl = getenv("NL_FUSED_SPIN_LIMIT")) e->spin_limit = atoi(sl);   // knob (tests): 0 makes every exchange give up
            if (const char *pf = getenv("NL_PREFETCH")) e->prefetch = atoi(pf);              // knob (A/B): bits of nl_engine::prefetch
        }
        const char *fa = getenv("NL_FUSED_ATTN");          // knob (tests, tools): 0 keeps the five-launch plan only, 2 forces mode 2
        const int want = fa ? atoi(fa) : -1;
        const bool base_ok = !(c.flags & NL_FLAG_LOCAL_GROUP) && want != 0 && e->hd == 64;
        bool ok1 = base_ok && want != 2 && e->G == 1 && !e->force_tp_plan && c.dim % (BLK_MEMBERS * TR) == 0 && c.n_layers < 255;
        // (an RCCL plan carries the all-reduced partial into the next projection's prologue, which the fused launch lacks)
        bool ok2 = base_ok && c.n_layers < 255 && ((e->G == 1 && !e->force_tp_plan) || e->p2p.on);
        // mode 3 (nl_tp.h): a tensor-parallel rank's layer as two launches -- ranks of a push group, or the shards of an
        // in-process group that asked for it (NL_FLAG_GROUP_FUSED: the same kernels, the group adds the partials itself)
        const bool group_fused = (c.flags & NL_FLAG_LOCAL_GROUP) && (c.flags & NL_FLAG_GROUP_FUSED);
        // wide_ffn_kernel (nl_tp.h) for this handle's (shard of the) feed-forward half: one workgroup per W_down tile, all resident
        // together.  NL_WIDE_FFN: 0 = never, 2 = also when its grid does not fill the chip (tests at small shapes)
        auto wide_ffn_ok = [&]() {
            const nl_engine::Layer &L0 = e->layers[0];
            const char *wf = getenv("NL_WIDE_FFN");
            const int wfv = wf ? atoi(wf) : 1;
            const int ngroups = (L0.gate.npairs + KL - 1) / KL, dgroups = (L0.down.npairs + KL - 1) / KL;
            bool okw = wfv != 0 && ngroups <= 16 && dgroups <= 48 && L0.gate.ntiles * GPT <= 5 * TP_THREADS && L0.down.ntiles <= e->num_cus &&
                       (L0.gate.ntiles + L0.down.ntiles - 1) / L0.down.ntiles <= 4 &&          // rounds are compile-time: 1 .. 4
                       // its grid is one workgroup per W_down tile: only a grid that fills the chip streams as fast as the GEMVs'
                       // (goldie, 96 tiles: 10.2 us against 4.6 + 4.0)
                       (wfv == 2 || L0.down.ntiles * 4 >= e->num_cus * 3);
            for (const auto &L : e->layers)
                okw = okw && L.gate.wtype == L.qkv.wtype && L.up.wtype == L.qkv.wtype && L.down.wtype == L.qkv.wtype &&
                      (L.qkv.wtype == WT_Q8_0 || L.qkv.wtype == WT_Q4_0) &&
                      L.gate.ntiles == L0.gate.ntiles && L.gate.npairs == L0.gate.npairs && L.up.ntiles == L0.gate.ntiles &&
                      L.down.npairs == L0.down.npairs && L.down.ntiles == L0.down.ntiles;
            e->wide_nf = (ngroups + 7) / 8;
            e->wide_ngc = (dgroups + 15) / 16 <= 1 ? 1 : 3;       // (two groups per wavefront run as three, the third masked)
            return okw && wide_ffn_lds_bytes(e->wide_nf, L0.down.npairs) <= (size_t)160 * 1024;
        };
        const char *tf = getenv("NL_TP_FUSED");            // knob (tests, tools): 0 keeps a rank on the projection + attention plan
        bool ok3 = want != 0 && want != 2 && !(tf && atoi(tf) == 0) && e->hd == 64 && e->G > 1 && c.n_layers < 127 &&
                   ((e->p2p.on && !(c.flags & NL_FLAG_LOCAL_GROUP)) || group_fused);
        for (const auto &L : e->layers) {
            ok1 = ok1 && L.wo_head.ready && L.wo_head.wtype == L.qkv.wtype && (L.qkv.wtype == WT_Q8_0 || L.qkv.wtype == WT_Q4_0);
            ok2 = ok2 && (L.qkv.wtype == WT_Q8_0 || L.qkv.wtype == WT_Q4_0);
            ok3 = ok3 && (L.qkv.wtype == WT_Q8_0 || L.qkv.wtype == WT_Q4_0) && L.wo.wtype == L.qkv.wtype &&
                  L.gate.wtype == L.qkv.wtype && L.up.wtype == L.qkv.wtype && L.down.wtype == L.qkv.wtype;
        }
        if ((ok2 && !ok1) || ok3) {
            // tiles per workgroup: as few as keeps every workgroup of the launch resident at once (one per compute unit)
            const int NT = (e->gqa + 2) * 4, ngroups = ((c.dim + PAIR - 1) / PAIR + KL - 1) / KL;
            int tpm = 1;
            const int cu_budget = e->num_cus - e->num_cus / 8;   // every workgroup of the launch resident at once, with a margin
            while (tpm <= 4 && e->KVs * NT / tpm > cu_budget) tpm *= 2;
            const bool geo = tpm <= 4 && NT % tpm == 0 && NT / tpm >= e->gqa && ngroups <= 2 * (16 / tpm);
            ok2 = ok2 && geo;
            ok3 = ok3 && geo;
            e->grp_tpm = tpm;
            // Modes 3 and 4 launch max(grp_grid, WO grid) blocks and, unlike mode 2, a block that holds no projection tile but
            // owns WO rows STAYS (it spins on the heads' outputs); the other tile-less blocks leave at once.  Every block that
            // stays -- live (cluster < KVs) or WO-owning (b * wo_tpw < WO tiles) -- must be resident together: a live block that
            // cannot be dispatched while WO owners wait for it would stall the launch into the fallback (stayers_fit below).
            e->grp_grid_fits = geo;      // (refined below, per mode, once the WO share of a block is known)
        }
        if (ok3) {
            nl_engine::TpGeom &t = e->tpg;
            const nl_engine::Layer &L0 = e->layers[0];
            // WO: EVERY block of the attention launch's grid owns tpw tiles of this rank's column slice; a row's 256-column
            // groups (a power of two <= 16) and the block's tiles share its 16 wavefronts
            const int wo_groups = (L0.wo.npairs + KL - 1) / KL;
            int gshift = 0;
            while ((1 << gshift) < wo_groups) gshift++;
            const int agrid = grp_grid(e->KVs, (e->gqa + 2) * 4 / e->grp_tpm);
            int tpw = 1;
            while (tpw < 16 && (long long)agrid * tpw < L0.wo.ntiles) tpw *= 2;
            ok3 = gshift <= 4 && (tpw << gshift) <= 16 && (long long)agrid * tpw >= L0.wo.ntiles && L0.wo.npairs <= 64;
            ok3 = ok3 && stayers_fit(e, (e->gqa + 2) * 4 / e->grp_tpm, agrid, tpw, L0.wo.ntiles);
            t.wo_gshift = gshift;
            t.wo_tpw = tpw;
            ok3 = ok3 && e->Hs * 4 * GPT <= 2 * TP_THREADS;  // at most two attention-output granules per thread
            // feed-forward: blocks [0, n_prod) project one gate / up tile each while that grid stays resident (one 1024-thread
            // workgroup per compute unit), else a gate + up tile pair each; EVERY block b < ceil(D / 16) owns W_down tile b
            const int ngroups = (L0.gate.npairs + KL - 1) / KL, dgroups = (L0.down.npairs + KL - 1) / KL;
            const int budget = e->num_cus - e->num_cus / 16;
            const char *pk = getenv("NL_TP_PAIR");     // developer knob (tools/, tests)
            t.pair = pk ? (atoi(pk) != 0) : (2 * L0.gate.ntiles > budget);
            t.n_prod = t.pair ? L0.gate.ntiles : 2 * L0.gate.ntiles;
            // W_down: blocks [0, n_cons) own ct tiles each (every consumer gathers all of h: fewer consumers = less fabric
            // traffic per gather, more tiles per consumer = more vector work per compute unit; 2 measured best at tp 8)
            const char *ck = getenv("NL_TP_CT");      // developer knob (tools/)
            int ct = ck ? atoi(ck) : (L0.down.npairs <= 24 ? 2 : 1);   // (tp 8 of the 7.9B tier: 2; tp 4: 1 -- measured, tools/r4_tp3.sh)
            if (ct != 1 && ct != 2 && ct != 4 && ct != 8 && ct != 16) ct = 2;
            while (ct > 1 && (dgroups + 16 / ct - 1) / (16 / ct) > 2) ct /= 2;
            t.ct_shift = ct == 1 ? 0 : ct == 2 ? 1 : ct == 4 ? 2 : ct == 8 ? 3 : 4;
            t.n_cons = (L0.down.ntiles + ct - 1) / ct;
            t.grid = std::max(t.n_prod, t.n_cons);
            t.nf = (ngroups + (t.pair ? 8 : 16) - 1) / (t.pair ? 8 : 16);
            t.ngc = (dgroups + 16 / ct - 1) / (16 / ct);
            t.nr = (L0.gate.ntiles * GPT + TP_THREADS - 1) / TP_THREADS;
            const bool tpffn = t.ngc <= 2 && t.nf <= 2 && t.nr <= 2 && t.n_prod <= budget && L0.down.npairs <= 128;
            const char *tw = getenv("NL_TP_WIDE");               // developer knob: 1 = prefer wide_ffn_kernel wherever it is eligible
            e->wide_ffn = ok3 && (!tpffn || (tw && atoi(tw) != 0)) && wide_ffn_ok();   // (a shard too large for one-tile producers: tp 2 of the 7.9B tier)
            ok3 = ok3 && (tpffn || e->wide_ffn);
            for (const auto &L : e->layers)   // (every layer has the shapes of layer 0: checked, not assumed)
                ok3 = ok3 && L.wo.npairs == L0.wo.npairs && L.gate.ntiles == L0.gate.ntiles && L.down.npairs == L0.down.npairs &&
                      L.down.ntiles == L0.down.ntiles && L.gate.npairs == L0.gate.npairs;
        }
        // mode 1 launches one 768-thread workgroup per (head, member) and one per (slice, member): they only make progress
        // together, so both grids must fit on the device's compute units at once
        if (ok1 && (blk_grid(e->Hs) > e->num_cus || ffn_grid(e->Is / FFN_SLICE) > e->num_cus)) ok1 = false;
        // mode 4: ONE GPU, wide tier -- mode 2's launch with WO behind a second exchange (nl_tp.h's attention half with no
        // all-reduce at all: every block that owns WO rows gathers the heads' outputs and stores x = resid + WO row itself).
        // Blocks beyond the projection's grid hold WO rows only; they wait for nobody but the runners, so only the projection's
        // grid has to be resident together (as in mode 2).  NL_ATTN_WO=0 keeps mode 2.
        bool ok4 = false;
        {
            const char *aw = getenv("NL_ATTN_WO");
            ok4 = ok2 && !ok1 && !ok3 && want != 2 && e->G == 1 && !e->p2p.on && !(aw && atoi(aw) == 0) && c.n_layers < 127 && e->grp_grid_fits;
            if (ok4) {
                nl_engine::TpGeom &t = e->tpg;
                const nl_engine::Layer &L0 = e->layers[0];
                const int wo_groups = (L0.wo.npairs + KL - 1) / KL;
                int gshift = 0;
                while ((1 << gshift) < wo_groups) gshift++;
                ok4 = gshift <= 4 && L0.wo.npairs <= 64 && e->Hs * 4 * GPT <= 2 * TP_THREADS;
                t.wo_gshift = gshift;
                t.wo_tpw = 16 >> std::min(gshift, 4);
                {
                    const int members = (e->gqa + 2) * 4 / e->grp_tpm;
                    const int grid4 = std::max(grp_grid(e->KVs, members), (L0.wo.ntiles + t.wo_tpw - 1) / t.wo_tpw);
                    ok4 = ok4 && stayers_fit(e, members, grid4, t.wo_tpw, L0.wo.ntiles);
                }
                for (const auto &L : e->layers)
                    ok4 = ok4 && (L.qkv.wtype == WT_Q8_0 || L.qkv.wtype == WT_Q4_0) && L.wo.wtype == L.qkv.wtype &&
                          L.wo.npairs == L0.wo.npairs && L.wo.ntiles == L0.wo.ntiles;
            }
        }
        e->fused_mode = ok1 ? 1 : ok3 ? 3 : ok4 ? 4 : ok2 ? 2 : 0;
        e->fused = e->fused_mode != 0;
        if (e->fused_mode != 3) e->wide_ffn = e->fused_mode == 4 && wide_ffn_ok();     // the feed-forward half as one launch too
        const char *fm = getenv("NL_FUSED_MAX_POS");
        // tools/fused_limit.py: the per-head blocks (mode 1) win up to ~500 (nano) / ~600 (mini) positions, the projection +
        // attention launch of the wide tiers (mode 2) up to ~390 (big)
        // (mode 4, tools/fused_limit.py on big, profiles/r04_big_fused_limit.log: with 256-position passes the two-launch layers
        //  stay ahead of the five-launch plan through three passes -- 1.69 against 1.85 ms at position 470, 1.81 against 1.86 at
        //  600 -- and fall behind in the fourth, 1.98 against 1.87 at 900)
        //  (mode 3 runs the same attention half: 512 = two of its passes, where 384 was three of the old 128-position ones)
        //  (round 5, the same tool with the feed-forward launch on the matrix pipe and its first round warm: 1.76 against 1.85 ms at
        //  position 800, 1.80 against 1.86 at 980 -- the two-launch layers now stay ahead through all four passes)
        e->fused_max_pos = fm ? atoi(fm) : e->fused_mode == 4 ? (e->wide_ffn ? 1024 : 768) : (e->fused_mode == 1 || e->fused_mode == 3) ? 512 : 384;
        {
            // long contexts on one GPU (mode 4): a head's 256-position passes are shared by its runner and three blocks that are not
            // runners -- two of its own kv group (members G .. 3G - 1) and one of the blocks past the kv groups (nl_tp.h) --, up to
            // two passes each: the two-launch layers then serve every position of the reference's context (2048, go/model.go:145-148)
            const int members = (e->gqa + 2) * 4 / e->grp_tpm, lgrid = grp_grid(e->KVs, members);
            const int grid4 = std::max(lgrid, (e->layers[0].wo.ntiles + e->tpg.wo_tpw - 1) / e->tpg.wo_tpw);
            const char *hk = getenv("NL_ATTN_HELPERS");
            e->attn_helpers = e->fused_mode == 4 && !(hk && atoi(hk) == 0) && !c.qk_norm && members - e->gqa >= 2 * e->gqa && grid4 - lgrid >= e->Hs &&
                              e->layers[0].qkv.wtype == WT_Q4_0;      // (the Q8_0 variant of that launch does not fit 128 registers: 278 spilled)
            if (!fm && e->attn_helpers && e->wide_ffn) e->fused_max_pos = 2 * TP_NCH_MAX * TP_PASS;
        }
        if (e->fused_mode == 3 || e->fused_mode == 4)      // passes a head takes inside the launch
            e->fused_max_pos = std::min(e->fused_max_pos, (e->attn_helpers ? 2 : 1) * TP_NCH_MAX * TP_PASS);
        {
            const char *ff = getenv("NL_FUSED_FFN");   // knob (tests, tools): 0 keeps gate/up and down as two launches
            bool okf = e->fused_mode == 1 && !(ff && atoi(ff) == 0);
            for (const auto &L : e->layers)
                okf = okf && L.dn_slice.ready && L.dn_slice.wtype == L.qkv.wtype && L.gate.wtype == L.qkv.wtype && L.up.wtype == L.qkv.wtype;
            e->ffn_fused = okf;
        }
        if (e->ffn_fused) {
            HIPCK(e, dalloc(&e->parts_ffn, (size_t)(e->Is / FFN_SLICE) * c.dim, &e->bytes_state));
            HIPCK(e, dalloc(&e->xchg_ffn, (size_t)e->Is, &e->bytes_state));
            HIPCK(e, hipMemset(e->xchg_ffn, 0, (size_t)e->Is * 8));
        }
        if (e->fused) {
            if (e->fused_mode == 1) HIPCK(e, dalloc(&e->parts, (size_t)e->Hs * c.dim, &e->bytes_state));
            const size_t nx = e->fused_mode == 1 ? (size_t)e->Hs * 192 : (size_t)e->KVs * (e->gqa + 2) * 64;
            HIPCK(e, dalloc(&e->xchg, nx, &e->bytes_state));
            HIPCK(e, hipMemset(e->xchg, 0, nx * 8));
            HIPCK(e, dalloc(&e->tick, (size_t)2, &e->bytes_state));
            HIPCK(e, hipMemset(e->tick, 0, 8));
            HIPCK(e, hipHostMalloc((void **)&e->h_status, sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
            *e->h_status = 0;
        }
        if (e->fused_mode == 3 || e->fused_mode == 4) {
            const size_t nq = (size_t)e->KVs * (e->gqa + 2) * 4 * GPT, no = (size_t)e->Hs * 4 * GPT, nh = (size_t)2 * e->layers[0].gate.ntiles * GPT;
            HIPCK(e, dalloc(&e->tp_xq, nq, &e->bytes_state));
            HIPCK(e, hipMemset(e->tp_xq, 0, nq * sizeof(u32x4)));
            HIPCK(e, dalloc(&e->tp_xo, no, &e->bytes_state));
            HIPCK(e, hipMemset(e->tp_xo, 0, no * sizeof(u32x4)));
            HIPCK(e, dalloc(&e->tp_xp, (size_t)e->Hs * 3 * 2 * 22, &e->bytes_state));
            HIPCK(e, hipMemset(e->tp_xp, 0, (size_t)e->Hs * 3 * 2 * 22 * sizeof(u32x4)));
            HIPCK(e, dalloc(&e->tp_hx, nh, &e->bytes_state));
            HIPCK(e, hipMemset(e->tp_hx, 0, nh * sizeof(u32x4)));
            // Q4_0 dot products of the two launches on the matrix pipe (nl_tp.h mf_*): matrices whose rows are whole 256-column
            // groups get the permuted second copy the quads of v_mfma_i32_4x4x4_16b_i8 need (+ their size in HBM)
            const char *mk = getenv("NL_MFMA_DOT");          // knob (A/B, tests): 0 keeps the vector-pipe dot products
            // (1, the default: the feed-forward launch only -- there a column group's digit image serves three rounds; in the attention
            //  launch it serves one tile and the conversion eats the gain: 13.5 against 13.3 us, profiles/r05_big_kernel_times_variants.log;
            //  2 switches that one on as well)
            bool ma = mk && atoi(mk) == 2 && e->Hs % 4 == 0, mf = !(mk && atoi(mk) == 0) && e->wide_ffn;
            for (const auto &L : e->layers) {
                ma = ma && L.qkv.wtype == WT_Q4_0 && L.qkv.npairs % KL == 0 && (L.qkv.npairs / KL) <= 16;
                mf = mf && L.gate.wtype == WT_Q4_0 && L.up.wtype == WT_Q4_0 && L.down.wtype == WT_Q4_0 && L.gate.npairs % KL == 0 && L.down.npairs % KL == 0;
            }
            auto permute = [&](PackedMat &m) -> int {
                if (m.q2) return NL_OK;
                HIPCK(e, arena_alloc(e, (void **)&m.q2, m.q_bytes));
                HIPCK(e, arena_alloc(e, (void **)&m.s2, m.s_bytes));
                e->bytes_weights += m.q_bytes + m.s_bytes;
                const long long groups = (long long)m.ntiles * (m.npairs / KL);
                hipLaunchKernelGGL(mf_permute_kernel, dim3((unsigned)((groups * 64 + 255) / 256)), dim3(256), 0, e->stream, reinterpret_cast<const uint4 *>(m.q), m.s,
                                   reinterpret_cast<uint4 *>(m.q2), m.s2, groups);
                HIPCK(e, hipGetLastError());
                return NL_OK;
            };
            for (auto &L : e->layers) {
                if (ma) { if (int prc = permute(L.qkv)) return prc; }
                if (mf) { if (int prc = permute(L.gate)) return prc; if (int prc = permute(L.up)) return prc; if (int prc = permute(L.down)) return prc; }
            }
            HIPCK(e, hipStreamSynchronize(e->stream));
            e->mf_attn = ma; e->mf_ffn = mf;
        }
    }
    const bool has_coll = (e->G > 1 || e->force_tp_plan) && !e->p2p.on;
    if (has_coll && !e->comm && !(c.flags & NL_FLAG_LOCAL_GROUP))
        return e->fail(NL_ERR_STATE, "collective plan needs nl_comm_init before nl_finalize");
    HIPCK(e, hipStreamSynchronize(e->stream));
    if (int rc = build_all(e)) return rc;
    if (int rc = pd_build(e)) return rc;
    e->finalized = true;
    return NL_OK;
}

// Gamma essence (go/gamma.go): embed[token] += gamma[token] for the listed tokens (go/model.go:503-505).
// indices: n token ids; values: [n][dim] float32, or raw IEEE binary16 when is_f16.  n == 0 clears it.
int nl_set_gamma(nl_handle e, const int32_t *indices, int n, const void *values, int is_f16) {
    if (e && e->grp) {
        // one rank after the other, as nl_finalize: the body allocates, frees and re-captures graphs, none of which may overlap
        // another thread's capture.  A rank that fails keeps its old table (the body only swaps on success); the ranks before it
        // are put back to "no gamma" and the call reports the error, so the ranks never decode with different embeddings.
        for (size_t r = 0; r < e->grp->members.size(); r++) {
            const int rc = nl_set_gamma(e->grp->members[r], indices, n, values, is_f16);
            if (rc != NL_OK) {
                e->err = "rank " + std::to_string(r) + ": " + e->grp->members[r]->err;
                for (size_t q = 0; q <= r; q++) (void)nl_set_gamma(e->grp->members[q], nullptr, 0, nullptr, 0);
                for (size_t q = r + 1; q < e->grp->members.size(); q++) (void)nl_set_gamma(e->grp->members[q], nullptr, 0, nullptr, 0);
                return rc;
            }
        }
        return NL_OK;
    }
    if (!e || n < 0 || (n > 0 && (!indices || !values))) return NL_ERR_INVALID;
    const nl_config &c = e->cfg;
    std::lock_guard<std::mutex> setup(g_setup_mu);      // (allocations, frees and graph captures below)
    HIPCK(e, hipSetDevice(e->dev));
    if (int qrc = pd_session_close(e)) return qrc;
    HIPCK(e, hipStreamSynchronize(e->stream));
    // Build and upload the new tables BEFORE touching the old ones: the plan closures and captured graphs hold the
    // old device pointers until they are rebuilt below, so no error path may leave them dangling.  Indices outside
    // [0, vocab) are dropped: the Go side keeps a map keyed by token id (go/gamma.go IndexMap) that such ids never hit.
    int *new_row = nullptr;
    float *new_val = nullptr;
    if (n > 0) {
        std::vector<int> rowmap(c.vocab, -1);
        for (int i = 0; i < n; i++)
            if (indices[i] >= 0 && indices[i] < c.vocab) rowmap[indices[i]] = i;  // later entries win, like the Go map build
        std::vector<float> vals((size_t)n * c.dim);
        if (is_f16) {
            const uint16_t *hp = (const uint16_t *)values;
            for (size_t i = 0; i < vals.size(); i++) vals[i] = __half2float(__ushort_as_half(hp[i]));
        } else memcpy(vals.data(), values, vals.size() * 4);
        hipError_t s = dalloc(&new_row, (size_t)c.vocab);
        if (s == hipSuccess) s = dalloc(&new_val, vals.size());
        if (s == hipSuccess) s = hipMemcpy(new_row, rowmap.data(), (size_t)c.vocab * 4, hipMemcpyHostToDevice);
        if (s == hipSuccess) s = hipMemcpy(new_val, vals.data(), vals.size() * 4, hipMemcpyHostToDevice);
        if (s != hipSuccess) {
            if (new_row) (void)hipFree(new_row);
            if (new_val) (void)hipFree(new_val);
            return e->fail(NL_ERR_HIP, "nl_set_gamma: %s", hipGetErrorString(s));   // old gamma still in force
        }
    }
    int *old_row = e->gamma_row;
    float *old_val = e->gamma_val;
    e->gamma_row = new_row;
    e->gamma_val = new_val;
    for (auto &G : e->bt_graphs) { (void)hipGraphExecDestroy(G.exec); (void)hipGraphDestroy(G.graph); }
    e->bt_graphs.clear();
    for (auto &sb : e->sub) {   // the sub-batch step graphs hold the old tables too
        for (auto &G : sb.graphs) { (void)hipGraphExecDestroy(G.exec); (void)hipGraphDestroy(G.graph); }
        sb.graphs.clear();
    }
    if (e->finalized) {  // the launch closures hold the old pointers: rebuild plans and graphs
        destroy_samp_graphs(e);
        if (build_all(e)) {   // the rebuilt eager plans are valid; only the graphs were lost
            for (auto &S : e->ps) destroy_graphs(S);
            e->use_graph = false;
        }
    }
    if (old_row) (void)hipFree(old_row);   // nothing references the old tables any more
    if (old_val) (void)hipFree(old_val);
    return NL_OK;
}

int nl_destroy(nl_handle e) {
    if (e && e->grp) {
        GroupCtl *g = e->grp;
        group_stop(g);
        for (nl_engine *m : g->members) nl_destroy(m);
        delete g;
        delete e;
        return NL_OK;
    }
    if (!e) return NL_OK;
    hipSetDevice(e->dev);
    (void)pd_session_close(e);
    pd_session_record(e, false);
    hipDeviceSynchronize();
    samp_free(e->sp);
    for (auto &S : e->ps) destroy_graphs(S);
    destroy_samp_graphs(e);
    for (auto &L : e->layers) {
        if (L.attn_norm) hipFree(L.attn_norm);
        if (L.ffn_norm) hipFree(L.ffn_norm);
        float *bs[] = {L.bq, L.bk, L.bv, L.bo};
        for (float *b : bs) if (b) hipFree(b);
    }
    pd_free(e);
    if (e->gamma_row) hipFree(e->gamma_row);
    if (e->gamma_val) hipFree(e->gamma_val);
    for (void *c : e->arena_chunks) hipFree(c);
    if (e->stage) hipFree(e->stage);
    auto batch_free = [](nl_engine::Batch &b) {
        void *bb[] = {b.x, b.qkv, b.q, b.g, b.u, b.logits, b.part_o, b.part_ml, b.tok /* | pos | stream */, b.ids, b.kpart, b.kpart2,
                      b.xfrag, b.xfrag2, b.ssq, b.nscale, b.kv16, b.tcos, b.tsin, b.tkv};
        for (void *p : bb) if (p) hipFree(p);
        if (b.h_meta) hipHostFree(b.h_meta);
    };
    batch_free(e->bt);
    for (auto &G : e->bt_graphs) { (void)hipGraphExecDestroy(G.exec); (void)hipGraphDestroy(G.graph); }
    for (auto &sb : e->sub) {
        for (auto &G : sb.graphs) { (void)hipGraphExecDestroy(G.exec); (void)hipGraphDestroy(G.graph); }
        batch_free(sb.bt);
        if (sb.done) (void)hipEventDestroy(sb.done);
        if (sb.st) (void)hipStreamDestroy(sb.st);
    }
    if (e->sub_fork) (void)hipEventDestroy(e->sub_fork);
    if (e->p2p.area) {
        if (e->p2p.on) e->logits = nullptr;   // lives inside the receive area (an export that was never imported allocated its own)
        for (int r = 0; r < 8; r++)
            if (e->p2p.opened[r]) (void)hipIpcCloseMemHandle(e->p2p.peer[r]);
        (void)hipFree(e->p2p.area);
        if (e->p2p.epoch) (void)hipFree(e->p2p.epoch);
    }
    void *bufs[] = {e->embd_raw, e->output_norm, e->rope_cos, e->rope_sin, e->x[0], e->x[1], e->qbuf, e->part_o,
                    e->part_ml, e->hb, e->ar, e->parts, e->xchg, e->tp_xq, e->tp_xo, e->tp_xp, e->tp_hx, e->tick, e->parts_ffn, e->xchg_ffn, e->samp_keep, e->logits, e->kcache, e->vcache, e->ctl, e->ids, e->result, e->amax_val,
                    e->amax_idx};
    for (void *b : bufs) if (b) hipFree(b);
    if (e->h_ctl) hipHostFree(e->h_ctl);
    if (e->h_logits) hipHostFree(e->h_logits);
    if (e->h_status) hipHostFree(e->h_status);
    if (e->h_ctl_ring) hipHostFree(e->h_ctl_ring);
    if (e->h_box) hipHostFree(e->h_box);
    if (e->h_done) hipHostFree(e->h_done);
    if (e->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm);
    if (e->ev0) hipEventDestroy(e->ev0);
    if (e->ev1) hipEventDestroy(e->ev1);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
    return NL_OK;
}

int nl_reset(nl_handle e, int stream) {
    if (e && e->grp) return group_run(e, [&](nl_engine *m, int) { return nl_reset(m, stream); });
    if (!e) return NL_ERR_INVALID;
    if (!e->finalized) return e->fail(NL_ERR_STATE, "reset before nl_finalize");
    if (stream < 0 || stream >= e->cfg.max_streams) return e->fail(NL_ERR_INVALID, "stream %d out of range", stream);
    // O(1): the 2*L*S*kvDim memset of go/model.go:623-631 is replaced by dropping the stream's high-water mark;
    // rows a later step could read without having rewritten them are cleared then (note_positions).
    e->hw[stream] = 0;
    return NL_OK;
}

float *nl_host_logits(nl_handle e) {
    if (e && e->grp) return e->grp->members.empty() ? nullptr : e->grp->members[0]->h_logits;
    return (e && e->finalized) ? e->h_logits : nullptr;
}

int nl_forward(nl_handle e, int stream, int token, int pos, float *logits_out) {
    if (e && e->grp) return group_run(e, [&](nl_engine *m, int r) { return nl_forward(m, stream, token, pos, r == 0 ? logits_out : nullptr); });
    if (!e) return NL_ERR_INVALID;
    int rc = check_step_args(e, stream, token, pos);
    if (rc) return rc;
    HIPCK(e, hipSetDevice(e->dev));
    {
        // the smallest tier: the resident session takes the token (nl_persist.h: no launch, no weight fetch, the LM-head units
        // store the logits into the pinned host buffer themselves)
        bool served = false;
        if ((rc = pd_session_step(e, stream, token, pos, logits_out != nullptr, nullptr, &served))) return rc;
        if (served) {
            if (logits_out && logits_out != e->h_logits) memcpy(logits_out, e->h_logits, (size_t)e->cfg.vocab * 4);
            return NL_OK;
        }
    }
    if (pd_usable(e, pos, 1)) {
        // ... or, sessions off: one token as a persistent launch of one step (weights to the registers, one pass)
        if ((rc = note_positions(e, stream, pos, 1))) return rc;
        const bool direct = logits_out && e->d_h_logits;
        if ((rc = pd_launch(e, stream, token, pos, 1, direct ? e->d_h_logits : nullptr))) return rc;
        if (logits_out && !direct) HIPCK(e, hipMemcpyAsync(e->h_logits, e->logits, (size_t)e->cfg.vocab * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCK(e, hipStreamSynchronize(e->stream));
        if (!pd_take_timeout(e)) {
            if (logits_out && logits_out != e->h_logits) memcpy(logits_out, e->h_logits, (size_t)e->cfg.vocab * 4);
            return NL_OK;
        }
    }
    for (int attempt = 0;; attempt++) {
        if ((rc = note_positions(e, stream, pos, 1))) return rc;
        // one GPU: the LM head stores the logits into the pinned buffer itself (ctl[CTL_HOSTOUT]); groups: a DMA behind the step
        const bool direct = logits_out && e->d_h_logits && e->G == 1 && !e->force_tp_plan && !e->p2p.on;
        if ((direct || !logits_out) && call_step(e, token, pos, stream, direct ? 1 : 0, nullptr, &rc)) {
            if (rc) return rc;       // (the call box: no control copy, no synchronise; the logits rows are in the pinned buffer)
        } else {
            if ((rc = set_ctl(e, token, pos, 0, stream, direct ? 1 : 0))) return rc;
            if ((rc = launch_step(e, pos))) return rc;
            if (logits_out && !direct)
                HIPCK(e, hipMemcpyAsync(e->h_logits, e->logits, (size_t)e->cfg.vocab * 4, hipMemcpyDeviceToHost, e->stream));
            HIPCK(e, hipStreamSynchronize(e->stream));
        }
        if (attempt == 0 && take_fused_timeout(e)) continue;   // redo on the general plan
        break;
    }
    if (int prc = p2p_check(e)) return prc;
    if (logits_out && logits_out != e->h_logits) memcpy(logits_out, e->h_logits, (size_t)e->cfg.vocab * 4);
    return NL_OK;
}

int nl_forward_argmax(nl_handle e, int stream, int token, int pos, int *next_id) {
    if (e && e->grp) {
        if (!next_id) return NL_ERR_INVALID;
        int ids[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int rc = group_run(e, [&](nl_engine *m, int r) { return nl_forward_argmax(m, stream, token, pos, &ids[r]); });
        *next_id = ids[0];
        return rc;
    }
    if (!e || !next_id) return NL_ERR_INVALID;
    int rc = check_step_args(e, stream, token, pos);
    if (rc) return rc;
    HIPCK(e, hipSetDevice(e->dev));
    {
        bool served = false;
        if ((rc = pd_session_step(e, stream, token, pos, false, next_id, &served))) return rc;
        if (served) return NL_OK;
    }
    if (pd_usable(e, pos, 1)) {          // (see nl_forward)
        if ((rc = note_positions(e, stream, pos, 1))) return rc;
        if ((rc = pd_launch(e, stream, token, pos, 1))) return rc;
        int *h_id = reinterpret_cast<int *>(e->h_logits + e->cfg.vocab);
        HIPCK(e, hipMemcpyAsync(h_id, e->ids, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCK(e, hipStreamSynchronize(e->stream));
        if (!pd_take_timeout(e)) { *next_id = *h_id; return NL_OK; }
    }
    for (int attempt = 0;; attempt++) {
        if ((rc = note_positions(e, stream, pos, 1))) return rc;
        int *h_id = reinterpret_cast<int *>(e->h_logits + e->cfg.vocab);
        if (call_step(e, token, pos, stream, 0, h_id, &rc)) {
            if (rc) return rc;
        } else {
            if ((rc = set_ctl(e, token, pos, 0, stream))) return rc;
            if ((rc = launch_step(e, pos))) return rc;
            HIPCK(e, hipMemcpyAsync(h_id, e->result, sizeof(int), hipMemcpyDeviceToHost, e->stream));
            HIPCK(e, hipStreamSynchronize(e->stream));
        }
        if (attempt == 0 && take_fused_timeout(e)) continue;   // redo on the general plan
        *next_id = *h_id;
        break;
    }
    if (int prc = p2p_check(e)) return prc;
    return NL_OK;
}

int nl_decode_greedy(nl_handle e, int stream, int token, int pos, int n_steps, int *ids_out, int *n_done) {
    if (e && e->grp) {
        if (!ids_out || n_steps < 0) return NL_ERR_INVALID;
        std::vector<std::vector<int>> ids(e->grp->members.size(), std::vector<int>((size_t)std::max(n_steps, 1)));
        int done[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int rc = group_run(e, [&](nl_engine *m, int r) { return nl_decode_greedy(m, stream, token, pos, n_steps, ids[r].data(), &done[r]); });
        if (rc == NL_OK) { memcpy(ids_out, ids[0].data(), (size_t)done[0] * sizeof(int)); if (n_done) *n_done = done[0]; }
        return rc;
    }
    if (!e || !ids_out || n_steps < 0) return NL_ERR_INVALID;
    int rc = check_step_args(e, stream, token, pos);
    if (rc) return rc;
    HIPCK(e, hipSetDevice(e->dev));
    if (int qrc = pd_session_close(e)) return qrc;
    int n = std::min(n_steps, e->cfg.seq_len - pos);
    n = std::min(n, e->ids_cap);
    if ((rc = note_positions(e, stream, pos, n))) return rc;
    int done = 0;
    if (n > 0 && pd_usable(e, pos, 1)) {
        // the smallest tier, short contexts: ONE persistent launch decodes the tokens below its position limit (nl_persist.h);
        // what lies beyond continues on the launch plans from the last id
        const int n1 = std::min(n, e->pd.max_pos - pos);
        if ((rc = pd_launch(e, stream, token, pos, n1))) return rc;
        HIPCK(e, hipMemcpyAsync(ids_out, e->ids, (size_t)n1 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCK(e, hipStreamSynchronize(e->stream));
        if (!pd_take_timeout(e)) done = n1;              // (a give-up: the whole chunk on the launch plans)
    }
    if (done < n) {
        const int tok = done ? ids_out[done - 1] : token, p0 = pos + done, m = n - done;
        for (int attempt = 0;; attempt++) {
            if ((rc = set_ctl(e, tok, p0, 1, stream))) return rc;
            int i = 0;
            for (; i + e->graph_steps <= m && e->graph_steps > 1; i += e->graph_steps) {
                nl_engine::PlanSet &S = pick_plan(e, p0 + i + e->graph_steps - 1);   // highest position of these steps
                if (!S.multi_exec) break;
                HIPCK(e, hipGraphLaunch(S.multi_exec, e->stream));
            }
            for (; i < m; i++)
                if ((rc = launch_step(e, p0 + i))) return rc;
            HIPCK(e, hipMemcpyAsync(ids_out + done, e->ids, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, e->stream));
            HIPCK(e, hipStreamSynchronize(e->stream));
            if (attempt == 0 && take_fused_timeout(e)) continue;   // the whole chain again, on the general plan
            break;
        }
    }
    if (int prc = p2p_check(e)) return prc;
    if (n_done) *n_done = n;
    return NL_OK;
}

namespace {

// ---- on-device sampling ----------------------------------------------------
// lane chunk of the select kernel: a multiple of 32 covering vocab / 1024 (vocab <= 131072)
inline int samp_chunk(int vocab) { return ((vocab + SAMP_THREADS - 1) / SAMP_THREADS + 31) / 32 * 32; }

hipError_t samp_alloc(SampScratch &s, int vocab, int n_uniforms) {
    hipError_t rc;
    if (samp_chunk(vocab) > 128) return hipErrorInvalidValue;   // vocab > 131072
    const int nblocks = (vocab + 255) / 256;
#define SA(ptr, count) if ((rc = hipMalloc((void **)&(ptr), (size_t)(count) * 4)) != hipSuccess) return rc
    SA(s.keys_in, SAMP_THREADS * samp_chunk(vocab));   // (whole lane chunks: the select kernels load past vocab and mask)
    if ((rc = hipMemset(s.keys_in, 0, (size_t)SAMP_THREADS * samp_chunk(vocab) * 4)) != hipSuccess) return rc;
    SA(s.scal, 4); SA(s.pmax, nblocks); SA(s.h1g, 2 * 2048);
    if ((rc = hipMemset(s.h1g, 0, 2048 * 8)) != hipSuccess) return rc;
    SA(s.uniforms, std::max(n_uniforms, 1)); SA(s.recent, SAMP_THREADS); SA(s.recent_n, 1);
#undef SA
    return hipSuccess;
}

void samp_free(SampScratch &s) {
    void *p[] = {s.keys_in, s.scal, s.pmax, s.h1g, s.uniforms, s.recent, s.recent_n};
    for (void *q : p) if (q) (void)hipFree(q);
    s = SampScratch{};
}

// one sampling decision on `logits` (device); advances ctl / ids / the recent window
hipError_t launch_sample(const SampScratch &s, float *logits, int vocab, const nl_sample_params &p, int *ctl, int *ids,
                         hipStream_t st, const EmbedParams *emb = nullptr) {
    SampleParams P{};
    if (emb) { P.embed = 1; P.emb = *emb; }
    P.logits = logits; P.vocab = vocab; P.temp = p.temperature; P.top_p = p.top_p; P.top_k = std::max(p.top_k, 1);
    P.rep_penalty = p.rep_penalty; P.recent = s.recent; P.recent_n = s.recent_n; P.rep_window = p.rep_window;
    P.uniforms = s.uniforms; P.ctl = ctl; P.ids = ids;
    P.keys_in = s.keys_in;
    P.scal = s.scal; P.pmax = s.pmax; P.nblocks = (vocab + 255) / 256;
    P.h1g = s.h1g; P.nblocks_pen = P.nblocks;
    // Three selections, none with a sort (nl_sample.h): temp <= 0 -> argmax of the penalised logits; top_p < 1 -> sampleTopP by
    // weighted radix selection (<= 32768 candidates in the registers of the one workgroup, up to 131072 streamed out of L2);
    // otherwise sampleTopK by a 2-bit-per-pass selection of the k-th largest logit + a bitonic network over the <= 1024 candidates
    P.radix = p.temperature > 0.f && p.top_p < 1.0f ? 1 : 0;
    hipLaunchKernelGGL(samp_penalty_kernel, dim3(P.nblocks), dim3(256), 0, st, P);
    if (P.radix) {
        hipLaunchKernelGGL(samp_prob_hist_kernel, dim3((vocab + SAMP_THREADS - 1) / SAMP_THREADS), dim3(SAMP_THREADS), 0, st, P);
        if (samp_chunk(vocab) == 32) hipLaunchKernelGGL(samp_select_radix_kernel<32>, dim3(1), dim3(SAMP_THREADS), 0, st, P);
        else hipLaunchKernelGGL(samp_select_radix_stream_kernel, dim3(1), dim3(SAMP_THREADS), 0, st, P);
    } else if (p.temperature > 0.f) {
        if (samp_chunk(vocab) == 32) hipLaunchKernelGGL(samp_topk_kernel<false>, dim3(1), dim3(SAMP_THREADS), 0, st, P);
        else hipLaunchKernelGGL(samp_topk_kernel<true>, dim3(1), dim3(SAMP_THREADS), 0, st, P);
    } else {
        hipLaunchKernelGGL(samp_argmax_select_kernel, dim3(1), dim3(SAMP_THREADS), 0, st, P);
    }
    return hipGetLastError();
}

int check_sample_params(nl_engine *e, const nl_sample_params *p) {
    auto bad = [&](const char *m) { return e ? e->fail(NL_ERR_INVALID, "%s", m) : (int)NL_ERR_INVALID; };
    if (!p) return bad("sample params missing");
    if (p->rep_window < 0 || p->rep_window > SAMP_THREADS) return bad("rep_window must be in [0, 1024] for on-device sampling");
    if (!(p->top_p > 0.f)) return bad("top_p must be > 0");
    if (p->top_k < 1 && p->top_p >= 1.f && p->temperature > 0.f) return bad("top_k must be >= 1");
    if (p->top_k > SAMP_THREADS && p->top_p >= 1.f && p->temperature > 0.f) return bad("top_k must be <= 1024 for on-device sampling (the host loop takes larger lists)");
    return NL_OK;
}

}  // namespace

int nl_sample_decode(nl_handle e, int stream, int pos, int n_steps, const nl_sample_params *p, const float *uniforms,
                     int *recent, int *n_recent, int *ids_out, int *n_done) {
    if (e && e->grp) {
        // every rank samples from the same gathered logits with the same uniforms: the same ids; rank 0's results are returned
        if (!ids_out || !uniforms || !recent || !n_recent || !p || n_steps < 0) return NL_ERR_INVALID;
        const size_t nr = e->grp->members.size(), cap = (size_t)std::max(p->rep_window, 1);
        std::vector<std::vector<int>> ids(nr, std::vector<int>((size_t)std::max(n_steps, 1))), rec(nr, std::vector<int>(cap));
        int nrec[8], done[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t r = 0; r < nr; r++) { nrec[r] = *n_recent; memcpy(rec[r].data(), recent, std::min((size_t)std::max(*n_recent, 0), cap) * sizeof(int)); }
        const int rc = group_run(e, [&](nl_engine *m, int r) { return nl_sample_decode(m, stream, pos, n_steps, p, uniforms, rec[r].data(), &nrec[r], ids[r].data(), &done[r]); });
        if (rc == NL_OK) {
            memcpy(ids_out, ids[0].data(), (size_t)done[0] * sizeof(int));
            memcpy(recent, rec[0].data(), (size_t)std::max(nrec[0], 0) * sizeof(int));
            *n_recent = nrec[0];
            if (n_done) *n_done = done[0];
        }
        return rc;
    }
    if (!e || !ids_out || !uniforms || !recent || !n_recent || n_steps < 0) return NL_ERR_INVALID;
    int rc = check_sample_params(e, p);
    if (rc) return rc;
    if (!e->finalized) return e->fail(NL_ERR_STATE, "sample before nl_finalize");
    if (stream < 0 || stream >= e->cfg.max_streams) return e->fail(NL_ERR_INVALID, "stream %d out of range", stream);
    if (pos < 1 || pos > e->cfg.seq_len) return e->fail(NL_ERR_INVALID, "pos %d out of range [1,%d]", pos, e->cfg.seq_len);
    if (*n_recent < 0 || *n_recent > p->rep_window) return e->fail(NL_ERR_INVALID, "n_recent %d exceeds rep_window %d", *n_recent, p->rep_window);
    HIPCK(e, hipSetDevice(e->dev));
    if (int qrc = pd_session_close(e)) return qrc;
    int n = std::min(n_steps, e->cfg.seq_len - pos);
    n = std::min(n, e->ids_cap);
    if (n_done) *n_done = std::max(n, 0);
    if (n <= 0) return NL_OK;
    if (!e->sp_ready || e->sp_uniforms_cap < n) {
        std::lock_guard<std::mutex> setup(g_setup_mu);
        HIPCK(e, hipStreamSynchronize(e->stream));
        destroy_samp_graphs(e);   // they hold the old scratch pointers
        samp_free(e->sp);
        e->sp_ready = false;
        e->sp_uniforms_cap = std::max(n, e->cfg.seq_len);
        HIPCK(e, samp_alloc(e->sp, e->cfg.vocab, e->sp_uniforms_cap));
        e->sp_ready = true;
    }
    const SampScratch &s = e->sp;
    const int n_recent_in = *n_recent;
    if (e->fused) {   // a redo after a fused-launch timeout restarts from the logits this call found (take_fused_timeout)
        if (!e->samp_keep) { std::lock_guard<std::mutex> setup(g_setup_mu); HIPCK(e, dalloc(&e->samp_keep, (size_t)e->cfg.vocab, &e->bytes_state)); }
        HIPCK(e, hipMemcpyAsync(e->samp_keep, e->logits, (size_t)e->cfg.vocab * 4, hipMemcpyDeviceToDevice, e->stream));
    }
    for (int attempt = 0;; attempt++) {
        if ((rc = note_positions(e, stream, pos, n))) return rc;
        // ctl: the select kernel writes token = sampled id and pos + 1; the forward that follows runs unchained
        // (its argmax does not advance the state).  ctl.pos is primed to pos - 1 so the first increment lands on pos.
        if ((rc = set_ctl(e, 0, pos - 1, 0, stream))) return rc;
        HIPCK(e, hipMemcpyAsync(s.uniforms, uniforms, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
        if (n_recent_in > 0) HIPCK(e, hipMemcpyAsync(s.recent, recent, (size_t)n_recent_in * 4, hipMemcpyHostToDevice, e->stream));
        *n_recent = n_recent_in;
        HIPCK(e, hipMemcpyAsync(s.recent_n, n_recent, 4, hipMemcpyHostToDevice, e->stream));
        int i = 0;
        while (e->ps[0].multi_exec && !e->samp_graph_failed && i + e->graph_steps <= n) {
            // {sampler kernels, plan} x graph_steps as one graph per launch plan (its kernel arguments include the sampling
            // parameters: re-captured when they change); the plan is chosen by the highest position of the 16 steps
            const int k = plan_index(e, pos + i + e->graph_steps - 1);
            if (!e->samp_graph_exec[k] || memcmp(&e->samp_graph_params[k], p, sizeof(*p)) != 0) {
                std::lock_guard<std::mutex> setup(g_setup_mu);
                if (e->samp_graph_exec[k]) { hipGraphExecDestroy(e->samp_graph_exec[k]); e->samp_graph_exec[k] = nullptr; }
                if (e->samp_graph[k]) { hipGraphDestroy(e->samp_graph[k]); e->samp_graph[k] = nullptr; }
                hipError_t cs = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal);
                int rc2 = NL_OK;
                // one GPU: the launch that picks the token also fetches its embedding row, and the plan's own embed and
                // argmax launches (the argmax of logits nobody reads) are left out of the graph -- two launches per token
                const bool tail = e->G == 1 && !e->p2p.on && !e->force_tp_plan && !getenv("NL_NO_SAMPLE_EMBED");
                for (int q = 0; cs == hipSuccess && q < e->graph_steps && !rc2; q++) {
                    if (launch_sample(s, e->logits, e->cfg.vocab, *p, e->ctl, e->ids, e->stream, tail ? &e->plan_embed : nullptr) != hipSuccess) rc2 = NL_ERR_HIP;
                    else if (!tail) rc2 = run_plan_eager(e, e->ps[k]);
                    else {
                        for (const Op &op : e->ps[k].ops) {
                            if (op.kind == K_EMBED || op.kind == K_ARGMAX) continue;
                            const hipError_t ls = op.fn(e->stream);
                            if (ls != hipSuccess) { rc2 = e->fail(NL_ERR_HIP, "launch %s: %s", kKindNames[op.kind], hipGetErrorString(ls)); break; }
                        }
                    }
                }
                hipGraph_t g = nullptr;
                if (cs == hipSuccess) cs = hipStreamEndCapture(e->stream, &g);
                if (cs == hipSuccess && !rc2 && g && hipGraphInstantiate(&e->samp_graph_exec[k], g, nullptr, nullptr, 0) == hipSuccess) {
                    e->samp_graph[k] = g;
                    e->samp_graph_params[k] = *p;
                } else {
                    if (g) hipGraphDestroy(g);
                    (void)hipGetLastError();
                    e->samp_graph_exec[k] = nullptr;
                    e->samp_graph_failed = true;      // (e.g. a library call that cannot be captured): eager launches below
                    break;
                }
            }
            HIPCK(e, hipGraphLaunch(e->samp_graph_exec[k], e->stream));
            i += e->graph_steps;
        }
        for (; i < n; i++) {
            HIPCK(e, launch_sample(s, e->logits, e->cfg.vocab, *p, e->ctl, e->ids, e->stream));
            if ((rc = launch_step(e, pos + i))) return rc;
        }
        HIPCK(e, hipMemcpyAsync(ids_out, e->ids, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCK(e, hipMemcpyAsync(n_recent, s.recent_n, 4, hipMemcpyDeviceToHost, e->stream));
        HIPCK(e, hipStreamSynchronize(e->stream));
        if (attempt == 0 && take_fused_timeout(e)) {
            HIPCK(e, hipMemcpyAsync(e->logits, e->samp_keep, (size_t)e->cfg.vocab * 4, hipMemcpyDeviceToDevice, e->stream));
            continue;
        }
        break;
    }
    if (int prc = p2p_check(e)) return prc;
    if (*n_recent > 0) HIPCK(e, hipMemcpy(recent, s.recent, (size_t)*n_recent * 4, hipMemcpyDeviceToHost));
    return NL_OK;
}

int nl_op_sample(int device, float *logits, int vocab, const nl_sample_params *p, float uniform, int *recent, int *n_recent,
                 int *picked) {
    if (!logits || vocab <= 0 || !recent || !n_recent || !picked) return NL_ERR_INVALID;
    int rc = check_sample_params(nullptr, p);
    if (rc) return rc;
    if (*n_recent < 0 || *n_recent > p->rep_window) return NL_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return NL_ERR_HIP;
    SampScratch s;
    float *d_logits = nullptr;
    int *d_ctl = nullptr, *d_ids = nullptr;
    rc = NL_ERR_HIP;
    do {
        if (samp_alloc(s, vocab, 1) != hipSuccess) break;
        if (hipMalloc((void **)&d_logits, (size_t)vocab * 4) != hipSuccess) break;
        if (hipMalloc((void **)&d_ctl, CTL_WORDS * 4) != hipSuccess || hipMalloc((void **)&d_ids, 16) != hipSuccess) break;
        if (hipMemset(d_ctl, 0, CTL_WORDS * 4) != hipSuccess) break;
        if (hipMemcpy(d_logits, logits, (size_t)vocab * 4, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(s.uniforms, &uniform, 4, hipMemcpyHostToDevice) != hipSuccess) break;
        if (*n_recent > 0 && hipMemcpy(s.recent, recent, (size_t)*n_recent * 4, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(s.recent_n, n_recent, 4, hipMemcpyHostToDevice) != hipSuccess) break;
        if (launch_sample(s, d_logits, vocab, *p, d_ctl, d_ids, (hipStream_t)0) != hipSuccess) break;
        if (hipDeviceSynchronize() != hipSuccess) break;
        if (hipMemcpy(picked, d_ids, 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        if (hipMemcpy(logits, d_logits, (size_t)vocab * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        if (hipMemcpy(n_recent, s.recent_n, 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        if (*n_recent > 0 && hipMemcpy(recent, s.recent, (size_t)*n_recent * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = NL_OK;
    } while (0);
    samp_free(s);
    if (d_logits) (void)hipFree(d_logits);
    if (d_ctl) (void)hipFree(d_ctl);
    if (d_ids) (void)hipFree(d_ids);
    return rc;
}

int nl_prefill(nl_handle e, int stream, const int *tokens, int n, int pos0, float *last_logits_out) {
    if (e && e->grp) return group_run(e, [&](nl_engine *m, int r) { return nl_prefill(m, stream, tokens, n, pos0, r == 0 ? last_logits_out : nullptr); });
    if (!e || !tokens || n < 0) return NL_ERR_INVALID;
    if (n == 0) return NL_OK;
    int rc = check_step_args(e, stream, tokens[0], pos0);
    if (rc) return rc;
    if (pos0 + n > e->cfg.seq_len) return e->fail(NL_ERR_INVALID, "prefill of %d tokens at pos %d exceeds seq_len %d", n, pos0, e->cfg.seq_len);
    for (int i = 0; i < n; i++)
        if (tokens[i] < 0 || tokens[i] >= e->cfg.vocab) return e->fail(NL_ERR_INVALID, "token %d out of range [0,%d)", tokens[i], e->cfg.vocab);
    HIPCK(e, hipSetDevice(e->dev));
    if (int qrc = pd_session_close(e)) return qrc;
    if ((rc = note_positions(e, stream, pos0, n))) return rc;
    if (n >= NL_BATCH_MIN && batch_supported(e)) {
        // multi-token path: 64-token tiles on the matrix cores; causality comes from each token's own pos
        { std::lock_guard<std::mutex> setup(g_setup_mu); if ((rc = batch_alloc(e, e->bt)) || (rc = dgemm_prepare(e))) return rc; }
        nl_engine::Batch &b = e->bt;
        for (int t0 = 0; t0 < n; t0 += b.cap) {
            const int m = std::min(b.cap, n - t0);
            HIPCK(e, hipStreamSynchronize(e->stream));  // h_meta is reused per tile
            for (int i = 0; i < m; i++) {
                b.h_meta[i] = tokens[t0 + i];
                b.h_meta[b.cap + i] = pos0 + t0 + i;
                b.h_meta[2 * b.cap + i] = stream;
            }
            const bool last = t0 + m == n;
            if ((rc = batched_step(e, b, e->stream, m, last ? 2 : 0, true))) return rc;
            if (last) {
                // keep the single-token state coherent: logits / argmax of the last token
                HIPCK(e, hipMemcpyAsync(e->logits, b.logits, (size_t)e->cfg.vocab * 4, hipMemcpyDeviceToDevice, e->stream));
                HIPCK(e, hipMemcpyAsync(e->result, b.ids, sizeof(int), hipMemcpyDeviceToDevice, e->stream));
            }
        }
    } else {
        for (int attempt = 0;; attempt++) {
            for (int i = 0; i < n; i++) {
                int *c = e->h_ctl_ring + (size_t)i * CTL_WORDS;
                c[CTL_TOKEN] = tokens[i]; c[CTL_POS] = pos0 + i; c[CTL_CHAIN] = 0; c[CTL_STEP] = 0; c[CTL_STREAM] = stream;
                HIPCK(e, hipMemcpyAsync(e->ctl, c, CTL_WORDS * sizeof(int), hipMemcpyHostToDevice, e->stream));
                if ((rc = launch_step(e, pos0 + i))) return rc;
            }
            HIPCK(e, hipStreamSynchronize(e->stream));
            if (attempt == 0 && take_fused_timeout(e)) continue;   // the prompt again, on the general plan
            break;
        }
    }
    if (last_logits_out)
        HIPCK(e, hipMemcpyAsync(e->h_logits, e->logits, (size_t)e->cfg.vocab * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCK(e, hipStreamSynchronize(e->stream));
    if (int prc = p2p_check(e)) return prc;
    if (last_logits_out && last_logits_out != e->h_logits) memcpy(last_logits_out, e->h_logits, (size_t)e->cfg.vocab * 4);
    return NL_OK;
}

int nl_forward_batch(nl_handle e, const int *streams, const int *tokens, const int *pos, int n, float *logits_out,
                     int *next_ids) {
    if (e && e->grp) {
        if (n < 0) return NL_ERR_INVALID;
        std::vector<std::vector<int>> ids(e->grp->members.size(), std::vector<int>((size_t)std::max(n, 1)));
        const int rc = group_run(e, [&](nl_engine *m, int r) { return nl_forward_batch(m, streams, tokens, pos, n, r == 0 ? logits_out : nullptr, ids[r].data()); });
        if (rc == NL_OK && next_ids) memcpy(next_ids, ids[0].data(), (size_t)n * sizeof(int));
        return rc;
    }
    if (!e || !streams || !tokens || !pos || n < 0) return NL_ERR_INVALID;
    if (n > e->cfg.max_streams) return e->fail(NL_ERR_INVALID, "batch of %d exceeds max_streams %d", n, e->cfg.max_streams);
    int rc;
    for (int i = 0; i < n; i++) {
        if ((rc = check_step_args(e, streams[i], tokens[i], pos[i]))) return rc;
        for (int j = 0; j < i; j++)
            if (streams[j] == streams[i]) return e->fail(NL_ERR_INVALID, "stream %d appears twice in one batch", streams[i]);
    }
    if (n == 0) return NL_OK;
    HIPCK(e, hipSetDevice(e->dev));
    if (int qrc = pd_session_close(e)) return qrc;
    for (int i = 0; i < n; i++)
        if ((rc = note_positions(e, streams[i], pos[i], 1))) return rc;
    if (n >= NL_BATCH_MIN && batch_supported(e)) {
        // The streams of a batch are independent: cut it into groups of >= 16 that step concurrently, each on its own HIP
        // stream with its own step buffers, so that one dependent chain's launch heads and memory round trips overlap
        // another's work (a 64-stream step of the 841M tier is ~224 launches of <= 144 workgroups on a 256-CU chip).  The
        // arithmetic per stream does not depend on the group it steps in: results are bitwise those of the one-step form.
        // (off unless NL_SUB_BATCHES > 1 -- measured slower, profiles/r04_subbatch_groups.log: the default is the chunked loop below)
        const int groups = e->sub_batches > 1 ? std::max((n + QG_TOK - 1) / QG_TOK, std::min(e->sub_batches, n / 16)) : 1;
        // (a step's launch list and kernel arguments also depend on the test knobs batched_step reads per step: all of them are
        //  part of the key of a cached step graph)
        auto batch_knob_sig = [] {
            const char *rk = getenv("NL_ROPE_IN_ATTN"), *fk = getenv("NL_FOLD_NORM"), *pk = getenv("NL_PREFILL_PRECISION"), *kk = getenv("NL_KV16_MIN_TOKENS"),
                       *dk = getenv("NL_DGEMM"), *dm = getenv("NL_DGEMM_MAX_TOKENS"), *hk = getenv("NL_DGEMM_HEAD");
            return 40503 * (hk ? atoi(hk) + 1 : 0) + (rk ? atoi(rk) + 1 : 0) + 3 * (fk ? atoi(fk) + 1 : 0) + 9 * (pk && !strcmp(pk, "fp16x1") ? 1 : 0) +
                   18 * (dk ? atoi(dk) + 1 : 0) + 54 * (dm ? (atoi(dm) & 0xff) + 1 : 0) + 13878 * (kk ? (atoi(kk) & 0xfff) + 1 : 0);
        };
        if (groups > 1 && groups <= 4) {
            if (!e->sub_fork) HIPCK(e, hipEventCreateWithFlags(&e->sub_fork, hipEventDisableTiming));
            HIPCK(e, hipEventRecord(e->sub_fork, e->stream));      // (the position bookkeeping above may have queued row clears)
            const int per = (n + groups - 1) / groups;
            // (a step's launch list and kernel arguments also depend on the test knobs batched_step reads per step: all of them
            //  are part of the key of a cached step graph)
            const int knob_sig = batch_knob_sig();
            for (int g = 0, t0 = 0; g < groups; g++, t0 += per) {
                const int m = std::min(per, n - t0);
                if (m <= 0) break;
                nl_engine::SubBatch &sb = e->sub[g];
                { std::lock_guard<std::mutex> setup(g_setup_mu); if ((rc = batch_alloc(e, sb.bt, QG_TOK)) || (rc = dgemm_prepare(e))) return rc; }
                if (!sb.st) {
                    HIPCK(e, hipStreamCreateWithFlags(&sb.st, hipStreamNonBlocking));
                    HIPCK(e, hipEventCreateWithFlags(&sb.done, hipEventDisableTiming));
                }
                nl_engine::Batch &b = sb.bt;
                int nsplit = 1;
                for (int i = 0; i < m; i++) {
                    b.h_meta[i] = tokens[t0 + i];
                    b.h_meta[b.cap + i] = pos[t0 + i];
                    b.h_meta[2 * b.cap + i] = streams[t0 + i];
                    nsplit = std::max(nsplit, pos[t0 + i] / ATT_CH + 1);
                }
                const int key_split = nsplit * 131072 + knob_sig;
                HIPCK(e, hipStreamWaitEvent(sb.st, e->sub_fork, 0));
                hipGraphExec_t exec = nullptr;
                for (auto &G : sb.graphs)
                    if (G.n == m && G.nsplit == key_split) exec = G.exec;
                if (!exec && e->use_graph) {
                    // the step reads its tokens / positions / streams from the pinned block behind h_meta (a copy node of the
                    // graph), so one graph serves every step of m streams with as many position splits
                    std::lock_guard<std::mutex> setup(g_setup_mu);      // (a capture must not overlap another thread's allocations)
                    hipGraph_t gr = nullptr;
                    if (hipStreamBeginCapture(sb.st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                        const int rc2 = batched_step(e, b, sb.st, m, 1);
                        const hipError_t ce = hipStreamEndCapture(sb.st, &gr);
                        if (!rc2 && ce == hipSuccess && gr && hipGraphInstantiate(&exec, gr, nullptr, nullptr, 0) == hipSuccess) {
                            sb.graphs.push_back({m, key_split, gr, exec});
                        } else {
                            if (gr) (void)hipGraphDestroy(gr);
                            (void)hipGetLastError();
                            exec = nullptr;
                        }
                    }
                    (void)hipGetLastError();
                }
                if (exec) HIPCK(e, hipGraphLaunch(exec, sb.st));
                else if ((rc = batched_step(e, b, sb.st, m, 1))) return rc;
                HIPCK(e, hipEventRecord(sb.done, sb.st));
                HIPCK(e, hipStreamWaitEvent(e->stream, sb.done, 0));     // later work on the engine's stream follows every group
            }
            // the read-backs only after EVERY group is queued: a copy into the caller's pageable memory blocks the host until
            // its stream has drained, and would serialise the groups
            for (int g = 0, t0 = 0; g < groups; g++, t0 += per) {
                const int m = std::min(per, n - t0);
                if (m <= 0) break;
                nl_engine::SubBatch &sb = e->sub[g];
                if (logits_out)
                    HIPCK(e, hipMemcpyAsync(logits_out + (size_t)t0 * e->cfg.vocab, sb.bt.logits, (size_t)m * e->cfg.vocab * 4,
                                            hipMemcpyDeviceToHost, sb.st));
                if (next_ids) HIPCK(e, hipMemcpyAsync(next_ids + t0, sb.bt.ids, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, sb.st));
            }
            for (int g = 0; g < groups; g++)
                if (e->sub[g].st) HIPCK(e, hipStreamSynchronize(e->sub[g].st));
            HIPCK(e, hipStreamSynchronize(e->stream));
            return NL_OK;
        }
        { std::lock_guard<std::mutex> setup(g_setup_mu); if ((rc = batch_alloc(e, e->bt)) || (rc = dgemm_prepare(e))) return rc; }
        nl_engine::Batch &b = e->bt;
        for (int t0 = 0; t0 < n; t0 += b.lm_cap) {
            const int m = std::min(b.lm_cap, n - t0);
            HIPCK(e, hipStreamSynchronize(e->stream));
            for (int i = 0; i < m; i++) {
                b.h_meta[i] = tokens[t0 + i];
                b.h_meta[b.cap + i] = pos[t0 + i];
                b.h_meta[2 * b.cap + i] = streams[t0 + i];
            }
            // The step as one hipGraph, cached by (streams, position splits, knobs): it reads its tokens / positions / streams from
            // the pinned block behind h_meta (a copy node of the graph), so one graph serves every step of that geometry.  The 150+
            // launches of a step are 3 us each when a host thread queues them one by one -- more than the kernels of a decode batch.
            hipGraphExec_t exec = nullptr;
            if (e->use_graph) {
                int nsplit = 1;
                for (int i = 0; i < m; i++) nsplit = std::max(nsplit, pos[t0 + i] / ATT_CH + 1);
                const int key_split = nsplit * 131072 + batch_knob_sig() % 131072;
                for (auto &G : e->bt_graphs)
                    if (G.n == m && G.nsplit == key_split) exec = G.exec;
                if (!exec) {
                    std::lock_guard<std::mutex> setup(g_setup_mu);      // (a capture must not overlap another thread's allocations)
                    hipGraph_t gr = nullptr;
                    if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                        const int rc2 = batched_step(e, b, e->stream, m, 1);
                        const hipError_t ce = hipStreamEndCapture(e->stream, &gr);
                        if (!rc2 && ce == hipSuccess && gr && hipGraphInstantiate(&exec, gr, nullptr, nullptr, 0) == hipSuccess) {
                            e->bt_graphs.push_back({m, key_split, gr, exec});
                        } else {
                            if (gr) (void)hipGraphDestroy(gr);
                            exec = nullptr;
                        }
                    }
                    (void)hipGetLastError();
                }
            }
            if (exec) HIPCK(e, hipGraphLaunch(exec, e->stream));
            else if ((rc = batched_step(e, b, e->stream, m, 1))) return rc;
            if (logits_out)
                HIPCK(e, hipMemcpyAsync(logits_out + (size_t)t0 * e->cfg.vocab, b.logits, (size_t)m * e->cfg.vocab * 4,
                                        hipMemcpyDeviceToHost, e->stream));
            if (next_ids) HIPCK(e, hipMemcpyAsync(next_ids + t0, b.ids, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, e->stream));
        }
        HIPCK(e, hipStreamSynchronize(e->stream));
        return NL_OK;
    }
    for (int attempt = 0;; attempt++) {
        for (int i = 0; i < n; i++) {
            int *c = e->h_ctl_ring + (size_t)i * CTL_WORDS;
            c[CTL_TOKEN] = tokens[i]; c[CTL_POS] = pos[i]; c[CTL_CHAIN] = 0; c[CTL_STEP] = 0; c[CTL_STREAM] = streams[i];
            HIPCK(e, hipMemcpyAsync(e->ctl, c, CTL_WORDS * sizeof(int), hipMemcpyHostToDevice, e->stream));
            if ((rc = launch_step(e, pos[i]))) return rc;
            if (logits_out)
                HIPCK(e, hipMemcpyAsync(logits_out + (size_t)i * e->cfg.vocab, e->logits, (size_t)e->cfg.vocab * 4,
                                        hipMemcpyDeviceToHost, e->stream));
            if (next_ids) HIPCK(e, hipMemcpyAsync(next_ids + i, e->result, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        }
        HIPCK(e, hipStreamSynchronize(e->stream));
        if (attempt == 0 && take_fused_timeout(e)) continue;   // every stream's step again, on the general plan
        break;
    }
    return NL_OK;
}

int nl_synchronize(nl_handle e) {
    if (e && e->grp) return group_run(e, [&](nl_engine *m, int) { return nl_synchronize(m); });
    if (!e) return NL_ERR_INVALID;
    HIPCK(e, hipSetDevice(e->dev));
    if (int qrc = pd_session_close(e)) return qrc;
    HIPCK(e, hipStreamSynchronize(e->stream));
    if (int prc = p2p_check(e)) return prc;
    return NL_OK;
}

int nl_timer_start(nl_handle e) {
    if (e && e->grp) return nl_timer_start(e->grp->members[0]);
    if (!e) return NL_ERR_INVALID;
    if (int qrc = pd_session_close(e)) return qrc;
    HIPCK(e, hipEventRecord(e->ev0, e->stream));
    return NL_OK;
}

int nl_timer_stop(nl_handle e, float *ms) {
    if (e && e->grp) return nl_timer_stop(e->grp->members[0], ms);
    if (!e || !ms) return NL_ERR_INVALID;
    if (int qrc = pd_session_close(e)) return qrc;
    HIPCK(e, hipEventRecord(e->ev1, e->stream));
    HIPCK(e, hipEventSynchronize(e->ev1));
    HIPCK(e, hipEventElapsedTime(ms, e->ev0, e->ev1));
    return NL_OK;
}

const char *nl_kernel_kind_name(int k) { return (k >= 0 && k < NL_NUM_KINDS) ? kKindNames[k] : ""; }

int nl_profile_forward(nl_handle e, int stream, int token, int pos, int iters, float *ms_out, int *calls_out) {
    if (e && e->grp) return e->fail(NL_ERR_UNSUPPORTED, "nl_profile_forward on a device group: profile one rank with nl_p2p_loopback instead");
    // One eager Forward to put valid data in every buffer, then every launch of the plan is replayed
    // `iters` times back to back between two HIP events on the engine's stream: per-launch time =
    // elapsed / iters (this includes the dependent-launch boundary, ~1-2 us, which a single short kernel
    // cannot be timed without; profiles/ holds the rocprofv3 pure-kernel durations).
    if (!e || !ms_out || !calls_out || iters <= 0) return NL_ERR_INVALID;
    int rc = check_step_args(e, stream, token, pos);
    if (rc) return rc;
    HIPCK(e, hipSetDevice(e->dev));
    if (int qrc = pd_session_close(e)) return qrc;
    for (int k = 0; k < NL_NUM_KINDS; k++) { ms_out[k] = 0.f; calls_out[k] = 0; }
    if ((rc = note_positions(e, stream, pos, 1))) return rc;
    if ((rc = set_ctl(e, token, pos, 0, stream))) return rc;
    const nl_engine::PlanSet *Sp = &pick_plan(e, pos);   // the plan a step at this position runs
    if ((rc = run_plan_eager(e, *Sp))) return rc;
    HIPCK(e, hipStreamSynchronize(e->stream));
    if (take_fused_timeout(e)) {
        Sp = &pick_plan(e, pos);
        if ((rc = run_plan_eager(e, *Sp))) return rc;
        HIPCK(e, hipStreamSynchronize(e->stream));
    }
    const nl_engine::PlanSet &S = *Sp;
    for (const Op &op : S.ops) {
        HIPCK(e, hipEventRecord(e->ev0, e->stream));
        for (int it = 0; it < iters; it++) {
            hipError_t s = op.fn(e->stream);
            if (s != hipSuccess) return e->fail(NL_ERR_HIP, "launch %s: %s", kKindNames[op.kind], hipGetErrorString(s));
            if ((rc = run_collective(e, op))) return rc;
        }
        HIPCK(e, hipEventRecord(e->ev1, e->stream));
        HIPCK(e, hipEventSynchronize(e->ev1));
        float ms = 0.f;
        HIPCK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
        ms_out[op.kind] += ms / (float)iters;
        calls_out[op.kind]++;
    }
    // the replays accumulated the residual epilogues `iters` times: x, logits and the K/V rows at `pos` of layers >= 1
    // are no longer a Forward's.  Positions >= pos of this stream count as unwritten from here on.
    e->hw[stream] = std::min(e->hw[stream], pos);
    return NL_OK;
}

int nl_plan_info(nl_handle e, int *fused_mode, int *fused_max_pos, int *launches_fused, int *launches_general) {
    if (e && e->grp) return nl_plan_info(e->grp->members[0], fused_mode, fused_max_pos, launches_fused, launches_general);
    if (!e) return NL_ERR_INVALID;
    if (!e->finalized) return e->fail(NL_ERR_STATE, "nl_plan_info before nl_finalize");
    if (fused_mode) *fused_mode = e->fused ? e->fused_mode : 0;
    if (fused_max_pos) *fused_max_pos = e->fused ? e->fused_max_pos : 0;
    if (launches_fused) *launches_fused = e->fused ? (int)e->ps[1].ops.size() : 0;
    if (launches_general) *launches_general = (int)e->ps[0].ops.size();
    return NL_OK;
}

int nl_persist_info(nl_handle e, int *ready, int *max_pos, long long *launches, long long *tokens) {
    if (e && e->grp) return nl_persist_info(e->grp->members[0], ready, max_pos, launches, tokens);
    if (!e) return NL_ERR_INVALID;
    if (ready) *ready = (e->pd.ready && !e->pd.retired) ? 1 : 0;
    if (max_pos) *max_pos = e->pd.ready ? e->pd.max_pos : 0;
    if (launches) *launches = e->pd.launches;
    if (tokens) *tokens = e->pd.tokens;
    return NL_OK;
}

int nl_memory_usage(nl_handle e, uint64_t *w, uint64_t *kv, uint64_t *st) {
    if (e && e->grp) {
        uint64_t a = 0, b = 0, c = 0;
        for (nl_engine *m : e->grp->members) { a += m->bytes_weights; b += m->bytes_kv; c += m->bytes_state; }
        if (w) *w = a; if (kv) *kv = b; if (st) *st = c;
        return NL_OK;
    }
    if (!e) return NL_ERR_INVALID;
    if (w) *w = e->bytes_weights;
    if (kv) *kv = e->bytes_kv;
    if (st) *st = e->bytes_state;
    return NL_OK;
}

int64_t nl_debug_read(nl_handle e, const char *which, int stream, float *out, int64_t max_floats) {
    if (e && e->grp) return nl_debug_read(e->grp->members[0], which, stream, out, max_floats);     // rank 0's buffers (the logits are the gathered ones)
    if (!e || !which || !out || !e->finalized) return NL_ERR_INVALID;
    const float *src = nullptr;
    int64_t n = 0;
    std::string w = which;
    if (w == "x") { src = e->x[0]; n = e->cfg.dim; }
    else if (w == "x1") { src = e->x[1]; n = e->cfg.dim; }
    else if (w == "q") { src = e->qbuf; n = (int64_t)e->Hs * e->hd; }
    else if (w == "hb") { src = e->hb; n = e->Is; }
    else if (w == "logits") { src = e->logits; n = e->cfg.vocab; }
    else if (w == "k_cache") { src = e->kcache + (long long)stream * e->kv_stream_stride; n = e->kv_stream_stride; }
    else if (w == "v_cache") { src = e->vcache + (long long)stream * e->kv_stream_stride; n = e->kv_stream_stride; }
    else if (w == "pd_dbg" && e->pd.dbg) { src = reinterpret_cast<const float *>(e->pd.dbg); n = 128; }   // 64 wall-clock stamps (int64) of nl_persist.h
    else return e->fail(NL_ERR_INVALID, "unknown debug buffer %s", which);
    n = std::min(n, max_floats);
    if (hipSetDevice(e->dev) == hipSuccess) (void)pd_session_close(e);
    if (hipSetDevice(e->dev) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess ||
        hipMemcpy(out, src, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return e->fail(NL_ERR_HIP, "debug read failed");
    return n;
}

// Phase timestamps (shader clock) of workgroup 0 for GEMV kind `kind` of layer 0; tools/ only.
int nl_debug_stamps(nl_handle e, int kind, long long *out /* 16 waves x 8 */) {
    if (e && e->grp) return e->fail(NL_ERR_UNSUPPORTED, "nl_debug_stamps on a device group: the leader holds no layers (profile one rank)");
    if (!e || !e->finalized || !out) return NL_ERR_INVALID;
    HIPCK(e, hipSetDevice(e->dev));
    if (int qrc = pd_session_close(e)) return qrc;
    long long *d = nullptr;
    HIPCK(e, hipMalloc((void **)&d, 128 * sizeof(long long)));
    HIPCK(e, hipMemset(d, 0, 128 * sizeof(long long)));
    const nl_config &c = e->cfg;
    nl_engine::Layer &L = e->layers[0];
    hipError_t s = hipErrorInvalidValue;
    if (kind == K_GATEUP) {
        GemvParams P = base_params(e, L.gate, 2);
        P.q1 = L.up.q; P.s1 = L.up.s; P.x = e->x[0]; P.normw = L.ffn_norm; P.out = e->hb; P.dbg = d;
        s = launch_gemv_t<PRO_NORM, EPI_SWIGLU>(L.gate.wtype, P, e->stream);
    } else if (kind == K_DOWN) {
        GemvParams P = base_params(e, L.down);
        P.x = e->hb; P.out = e->x[1]; P.resid = e->x[1]; P.dbg = d;
        s = launch_gemv_t<PRO_PLAIN, EPI_RESID>(L.down.wtype, P, e->stream);
    } else if (kind == K_FFNBLOCK && e->fused && e->ffn_fused) {
        for (const Op &op : e->ps[1].ops)
            if (op.kind == K_FFNBLOCK) {
                e->dbg_block = d;
                s = op.fn(e->stream);
                e->dbg_block = nullptr;
                break;
            }
    } else if (kind == K_ATTNBLOCK && e->fused) {
        // layer 0's block launch as the plan holds it, with phase stamps (position / stream as ctl has them)
        for (const Op &op : e->ps[1].ops)
            if (op.kind == K_ATTNBLOCK) {
                e->dbg_block = d;
                s = op.fn(e->stream);
                e->dbg_block = nullptr;
                break;
            }
    }
    (void)c;
    if (s != hipSuccess) { hipFree(d); return e->fail(NL_ERR_HIP, "debug launch: %s", hipGetErrorString(s)); }
    HIPCK(e, hipStreamSynchronize(e->stream));
    HIPCK(e, hipMemcpy(out, d, 128 * sizeof(long long), hipMemcpyDeviceToHost));
    hipFree(d);
    return NL_OK;
}

// ---- op-level entry points ---------------------------------------------------

int nl_op_matmul(int device, uint32_t type, const void *w, uint64_t nbytes, const float *x, float *out, int rows,
                 int cols) {
    if (!w || !x || !out || rows <= 0 || cols <= 0 || cols % 32) return NL_ERR_INVALID;
    if (!type_supported(type)) return NL_ERR_UNSUPPORTED;  // matmulDispatch default arm, go/model.go:383-385
    if (is_kquant(type) && cols % 256) return NL_ERR_INVALID;
    if (raw_bytes(type, (uint64_t)rows * cols) != nbytes) return NL_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return NL_ERR_HIP;
    nl_engine tmp;
    tmp.hd = 64;
    hipStream_t st;
    if (hipStreamCreate(&st) != hipSuccess) return NL_ERR_HIP;
    tmp.stream = st;
    PackedMat m;
    uint8_t *d_raw = nullptr;
    float *d_x = nullptr, *d_out = nullptr;
    int rc = NL_ERR_HIP;
    do {
        if (alloc_packed(&tmp, m, (int)type, (rows + TR - 1) / TR, rows, cols) != hipSuccess) break;
        if (hipMalloc((void **)&d_raw, nbytes) != hipSuccess) break;
        if (hipMalloc((void **)&d_x, (size_t)cols * 4) != hipSuccess) break;
        if (hipMalloc((void **)&d_out, (size_t)rows * 4) != hipSuccess) break;
        if (hipMemcpyAsync(d_raw, w, nbytes, hipMemcpyHostToDevice, st) != hipSuccess) break;
        if (hipMemcpyAsync(d_x, x, (size_t)cols * 4, hipMemcpyHostToDevice, st) != hipSuccess) break;
        if (repack(&tmp, m, d_raw, (int)type, cols, 0, rows, 0, cols, 0, m.ntiles, ROWMAP_IDENT) != hipSuccess) break;
        GemvParams P{};
        P.q0 = m.q; P.s0 = m.s; P.rows = rows; P.cols = cols; P.npairs = m.npairs; P.ntiles = m.ntiles;
        if (const char *v = getenv("NL_TW")) tmp.tw_override = atoi(v);
        if (const char *v = getenv("NL_KW")) tmp.kw_override = atoi(v);
        choose_geometry(&tmp, m, P.tw, P.kw);
        P.x = d_x; P.out = d_out;
        if (launch_gemv_t<PRO_PLAIN, EPI_STORE>(m.wtype, P, st) != hipSuccess) break;
        if (hipMemcpyAsync(out, d_out, (size_t)rows * 4, hipMemcpyDeviceToHost, st) != hipSuccess) break;
        if (hipStreamSynchronize(st) != hipSuccess) break;
        rc = NL_OK;
    } while (0);
    for (void *c : tmp.arena_chunks) hipFree(c);
    if (d_raw) hipFree(d_raw);
    if (d_x) hipFree(d_x);
    if (d_out) hipFree(d_out);
    hipStreamDestroy(st);
    tmp.stream = nullptr;
    return rc;
}



// Multi-token matmul through the MFMA path: out[n][rows] = W @ x[n] for n_tokens vectors (host in/out).
int nl_op_matmul_batch(int device, uint32_t type, const void *w, uint64_t nbytes, const float *x, float *out,
                       int rows, int cols, int n_tokens) {
    if (!w || !x || !out || rows <= 0 || cols <= 0 || cols % 32 || n_tokens <= 0) return NL_ERR_INVALID;
    if (device_type((int)type) != WT_Q4_0 && device_type((int)type) != WT_Q8_0 && type != WT_F16) return NL_ERR_UNSUPPORTED;
    if (raw_bytes(type, (uint64_t)rows * cols) != nbytes) return NL_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return NL_ERR_HIP;
    nl_engine tmp;
    tmp.hd = 64;
    hipStream_t st;
    if (hipStreamCreate(&st) != hipSuccess) return NL_ERR_HIP;
    tmp.stream = st;
    PackedMat m;
    uint8_t *d_raw = nullptr;
    float *d_x = nullptr, *d_out = nullptr;
    uint4 *d_xf = nullptr;
    int rc = NL_ERR_HIP;
    do {
        if (alloc_packed(&tmp, m, (int)type, (rows + TR - 1) / TR, rows, cols) != hipSuccess) break;
        if (hipMalloc((void **)&d_raw, nbytes) != hipSuccess) break;
        if (hipMalloc((void **)&d_x, (size_t)cols * n_tokens * 4) != hipSuccess) break;
        if (hipMalloc((void **)&d_out, (size_t)rows * n_tokens * 4) != hipSuccess) break;
        if (hipMemcpyAsync(d_raw, w, nbytes, hipMemcpyHostToDevice, st) != hipSuccess) break;
        if (hipMemcpyAsync(d_x, x, (size_t)cols * n_tokens * 4, hipMemcpyHostToDevice, st) != hipSuccess) break;
        if (repack(&tmp, m, d_raw, (int)type, cols, 0, rows, 0, cols, 0, m.ntiles, ROWMAP_IDENT) != hipSuccess) break;
        QGemmParams P{};
        P.q = m.q; P.s = m.s; P.rows = rows; P.cols = cols; P.npairs = m.npairs; P.ntiles = m.ntiles;
        if (hipMalloc((void **)&d_xf, xfrag_uint4(cols, n_tokens) * sizeof(uint4)) != hipSuccess) break;
        if (launch_xsplit(m.wtype, d_x, cols, cols, n_tokens, d_xf, st) != hipSuccess) break;
        P.xf = d_xf; P.n_tokens = n_tokens; P.out = d_out; P.ldo = rows;
        if (launch_qgemm(m.wtype, P, st) != hipSuccess) break;
        if (hipMemcpyAsync(out, d_out, (size_t)rows * n_tokens * 4, hipMemcpyDeviceToHost, st) != hipSuccess) break;
        if (hipStreamSynchronize(st) != hipSuccess) break;
        rc = NL_OK;
    } while (0);
    for (void *c : tmp.arena_chunks) hipFree(c);
    if (d_raw) hipFree(d_raw);
    if (d_xf) hipFree(d_xf);
    if (d_x) hipFree(d_x);
    if (d_out) hipFree(d_out);
    hipStreamDestroy(st);
    tmp.stream = nullptr;
    return rc;
}

namespace {
__global__ void rmsnorm_op_kernel(const float *x, const float *w, float eps, float *out, int n) {
    __shared__ double dred[16];
    double ss = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) ss += (double)x[i] * (double)x[i];
    ss = wave_sum_f64(ss);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = ss;
    __syncthreads();
    double tot = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) tot += dred[k];
    float inv = (float)(1.0 / sqrt(tot / (double)n + (double)eps));
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = (x[i] * inv) * w[i];
}
}  // namespace

int nl_op_rmsnorm(int device, const float *x, const float *w, float eps, float *out, int n) {
    if (!x || !w || !out || n <= 0) return NL_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return NL_ERR_HIP;
    float *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)n * 12) != hipSuccess) return NL_ERR_HIP;
    int rc = NL_ERR_HIP;
    do {
        if (hipMemcpy(d, x, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(d + n, w, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) break;
        hipLaunchKernelGGL(rmsnorm_op_kernel, dim3(1), dim3(256), 0, 0, d, d + n, eps, d + 2 * n, n);
        if (hipGetLastError() != hipSuccess) break;
        if (hipMemcpy(out, d + 2 * n, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = NL_OK;
    } while (0);
    hipFree(d);
    return rc;
}

namespace {
__global__ void exp_op_kernel(const float *x, float *out, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = exp_f64_as_f32(x[i]);
}
}  // namespace

// float32(exp(float64(x))) as the forward kernels compute it (exp_f64_as_f32, nl_kernels.h): parity tests only.
int nl_op_exp(int device, const float *x, float *out, int n) {
    if (!x || !out || n <= 0) return NL_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return NL_ERR_HIP;
    float *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)n * 8) != hipSuccess) return NL_ERR_HIP;
    int rc = NL_ERR_HIP;
    do {
        if (hipMemcpy(d, x, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) break;
        hipLaunchKernelGGL(exp_op_kernel, dim3(1024), dim3(256), 0, 0, d, d + n, n);
        if (hipGetLastError() != hipSuccess) break;
        if (hipMemcpy(out, d + n, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = NL_OK;
    } while (0);
    hipFree(d);
    return rc;
}

// Soak of the 16-byte granule form (nl_tp.h gran16_soak_kernel): pairs writers on `n` compute units with readers on other XCDs.
int nl_op_gran16_soak(int device, int n, unsigned iters, unsigned long long *torn, unsigned long long *seen) {
    if (!torn || !seen || n < 1 || n > 128 || iters < 1) return NL_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return NL_ERR_HIP;
    u32x4 *slots = nullptr;
    unsigned long long *cnt = nullptr;
    int rc = NL_ERR_HIP;
    do {
        if (hipMalloc((void **)&slots, (size_t)n * 256 * sizeof(u32x4)) != hipSuccess || hipMalloc((void **)&cnt, 16) != hipSuccess) break;
        if (hipMemset(slots, 0, (size_t)n * 256 * sizeof(u32x4)) != hipSuccess || hipMemset(cnt, 0, 16) != hipSuccess) break;
        hipLaunchKernelGGL(gran16_soak_kernel, dim3(2 * n), dim3(256), 0, 0, slots, n, iters, cnt, cnt + 1);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) break;
        unsigned long long h[2];
        if (hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost) != hipSuccess) break;
        *torn = h[0]; *seen = h[1];
        rc = NL_OK;
    } while (0);
    if (slots) hipFree(slots);
    if (cnt) hipFree(cnt);
    return rc;
}

// ---- in-process tensor-parallel group ----------------------------------------

int nl_group_forward(nl_handle *hs, int n, int stream, int token, int pos, float *logits_out) {
    if (!hs || n < 1 || n > 8) return NL_ERR_INVALID;
    nl_engine *e0 = hs[0];
    if (!e0) return NL_ERR_INVALID;
    for (int r = 0; r < n; r++) {
        if (!hs[r] || hs[r]->G != n || hs[r]->rank != r || !(hs[r]->cfg.flags & NL_FLAG_LOCAL_GROUP))
            return e0->fail(NL_ERR_INVALID, "nl_group_forward: shard %d is not rank %d of a local group of %d", r, r, n);
        int rc = check_step_args(hs[r], stream, token, pos);
        if (rc) return rc;
        if (hs[r]->dev != e0->dev) return e0->fail(NL_ERR_UNSUPPORTED, "local group across devices is not implemented");
    }
    HIPCK(e0, hipSetDevice(e0->dev));
    hipStream_t st = e0->stream;  // one stream serialises the whole group
    for (int attempt = 0;; attempt++) {
        // NL_FLAG_GROUP_FUSED shards step the two-launches-per-layer plan of a push-group rank (nl_tp.h) at short contexts;
        // its launches leave every shard's partial in `ar` and the group adds them in rank order (coll 3)
        int k = 1;
        for (int r = 0; r < n; r++)
            if (!(hs[r]->fused && hs[r]->fused_mode == 3 && pos < hs[r]->fused_max_pos)) k = 0;
        for (int r = 0; r < n; r++)
            if (hs[r]->ps[k].ops.size() != e0->ps[k].ops.size()) return e0->fail(NL_ERR_STATE, "shards disagree on the launch plan");
        for (int r = 0; r < n; r++) {
            nl_engine *e = hs[r];
            if (int rc = note_positions(e, stream, pos, 1, st)) return rc;
            e->h_ctl[CTL_TOKEN] = token; e->h_ctl[CTL_POS] = pos; e->h_ctl[CTL_CHAIN] = 0;
            e->h_ctl[CTL_STEP] = 0; e->h_ctl[CTL_STREAM] = stream;
            HIPCK(e0, hipMemcpyAsync(e->ctl, e->h_ctl, CTL_WORDS * sizeof(int), hipMemcpyHostToDevice, st));
        }
        for (size_t i = 0; i < e0->ps[k].ops.size(); i++) {
            for (int r = 0; r < n; r++) {
                hipError_t s = hs[r]->ps[k].ops[i].fn(st);
                if (s != hipSuccess) return e0->fail(NL_ERR_HIP, "group launch %s: %s", kKindNames[e0->ps[k].ops[i].kind], hipGetErrorString(s));
            }
            const Op &op = e0->ps[k].ops[i];
            if (op.coll == 1 || op.coll == 3) {
                PtrList8 pl{};
                for (int r = 0; r < n; r++) pl.p[r] = hs[r]->ps[k].ops[i].buf;
                int cnt = (int)op.count;
                hipLaunchKernelGGL(local_allreduce_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, pl, n, cnt);
                HIPCK(e0, hipGetLastError());
                if (op.coll == 3)
                    for (int r = 0; r < n; r++) {
                        const Op &o = hs[r]->ps[k].ops[i];
                        hipLaunchKernelGGL(tp_add_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, o.add_to, o.buf, cnt);
                        HIPCK(e0, hipGetLastError());
                    }
            } else if (op.coll == 2) {
                for (int src = 0; src < n; src++)
                    for (int dst = 0; dst < n; dst++)
                        if (src != dst)
                            HIPCK(e0, hipMemcpyAsync(hs[dst]->ps[k].ops[i].buf + (size_t)src * op.count,
                                                     hs[src]->ps[k].ops[i].buf + (size_t)src * op.count, op.count * 4,
                                                     hipMemcpyDeviceToDevice, st));
            }
        }
        if (logits_out)
            HIPCK(e0, hipMemcpyAsync(logits_out, e0->logits, (size_t)e0->cfg.vocab * 4, hipMemcpyDeviceToHost, st));
        HIPCK(e0, hipStreamSynchronize(st));
        bool redo = false;
        for (int r = 0; r < n; r++)
            if (hs[r]->h_status && *hs[r]->h_status) redo = true;
        if (attempt == 0 && redo) {          // a cluster exchange gave up in some shard: every shard retires the fused plan
            for (int r = 0; r < n; r++) {
                nl_engine *e = hs[r];
                if (e->h_status) *e->h_status = 0;
                if (e->tick) (void)hipMemsetAsync(e->tick + 1, 0, sizeof(unsigned), st);
                e->fused = false;
                e->fused_retired = true;
            }
            HIPCK(e0, hipStreamSynchronize(st));
            e0->fail(NL_OK, "warning: a fused-launch cluster exchange timed out in the shard group; the step was redone on the general plan");
            continue;
        }
        break;
    }
    return NL_OK;
}

// ---- tensor-parallel communicator -------------------------------------------

int nl_comm_get_unique_id(void *id_out) {
    if (!id_out) return NL_ERR_INVALID;
    std::string err;
    if (!g_rccl.load(err)) { g_create_error = err; return NL_ERR_COMM; }
    NcclId id;
    if (g_rccl.GetUniqueId(&id) != 0) { g_create_error = "ncclGetUniqueId failed"; return NL_ERR_COMM; }
    memcpy(id_out, &id, NL_COMM_ID_BYTES);
    return NL_OK;
}

int nl_comm_init(nl_handle e, const void *id) {
    if (e && e->grp) return e->fail(NL_ERR_UNSUPPORTED, "nl_comm_init on a device group: nl_create_group wires its ranks itself");
    if (!e || !id) return NL_ERR_INVALID;
    if (e->finalized) return e->fail(NL_ERR_STATE, "nl_comm_init after nl_finalize");
    if (e->G <= 1 && !e->force_tp_plan) return NL_OK;
    std::string err;
    if (!g_rccl.load(err)) return e->fail(NL_ERR_COMM, "%s", err.c_str());
    HIPCK(e, hipSetDevice(e->dev));
    NcclId nid;
    memcpy(&nid, id, NL_COMM_ID_BYTES);
    int rc = g_rccl.CommInitRank(&e->comm, e->G, nid, e->rank);
    if (rc != 0) return e->fail(NL_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return NL_OK;
}

// ---- push all-reduce between the ranks of one node (nl_p2p.h) ------------------

namespace {
// this rank's receive area of the push all-reduce (uncached; see nl_p2p_export), allocated once
int p2p_alloc_area(nl_engine *e) {
    HIPCK(e, hipSetDevice(e->dev));
    nl_engine::P2P &p = e->p2p;
    if (!p.area) {
        const size_t ar = (size_t)2 * e->G * e->cfg.dim * sizeof(u64);
        p.off_amax = (ar + 255) & ~(size_t)255;
        p.off_logits = (p.off_amax + (size_t)e->G * 4 * sizeof(u64) + 255) & ~(size_t)255;
        p.bytes = p.off_logits + (size_t)e->cfg.vocab * 4;
        // Written by the peers while this device reads it: the area must be UNCACHED, so that no L2 line of this device
        // can shadow a push (remote xGMI writes do not probe the local L2).  The tagged granules would survive a cached
        // area -- their polls time out -- but the logits all-gather stores plain floats: a stale line would be read
        // silently.  So there is no cached fallback: without an uncached, exportable allocation the push path is
        // refused and the caller (bench.py, load_llama_model) falls back to RCCL.  NL_P2P_CACHED=1 is a test knob for
        // ranks that share ONE device (one L2, nothing to shadow); it is rejected across devices at import.
        const bool test_cached = getenv("NL_P2P_CACHED") != nullptr;
        hipError_t s = test_cached ? hipMalloc(&p.area, p.bytes) : hipExtMallocWithFlags(&p.area, p.bytes, hipDeviceMallocUncached);
        if (s != hipSuccess) {
            (void)hipGetLastError();
            p.area = nullptr;
            return e->fail(NL_ERR_UNSUPPORTED, "push all-reduce needs an uncached receive area (hipExtMallocWithFlags: %s); use RCCL", hipGetErrorString(s));
        }
        p.uncached = !test_cached;
        HIPCK(e, hipMemset(p.area, 0, p.bytes));
        HIPCK(e, hipMalloc((void **)&p.epoch, 2 * sizeof(unsigned)));
        HIPCK(e, hipMemset(p.epoch, 0, 2 * sizeof(unsigned)));
        p.status = p.epoch + 1;
        const char *tm = getenv("NL_P2P_TIMEOUT_MS");
        p.timeout_ticks = (long long)(tm ? atoi(tm) : 10000) * 100000;   // wall_clock64: 100 MHz
        e->bytes_state += p.bytes;
    }
    return NL_OK;
}
}  // namespace

int nl_p2p_export(nl_handle e, void *handle_out) {
    if (e && e->grp) return e->fail(NL_ERR_UNSUPPORTED, "nl_p2p_export on a device group: nl_create_group wires its ranks itself");
    if (!e || !handle_out) return NL_ERR_INVALID;
    if (e->finalized) return e->fail(NL_ERR_STATE, "nl_p2p_export after nl_finalize");
    if (e->G < 2) return e->fail(NL_ERR_INVALID, "nl_p2p_export needs tp_size >= 2");
    if (e->G != 2 && e->G != 4 && e->G != 8) return e->fail(NL_ERR_UNSUPPORTED, "push all-reduce supports 2, 4 or 8 ranks");
    if (e->cfg.n_layers > 127) return e->fail(NL_ERR_UNSUPPORTED, "push all-reduce tags cover at most 127 layers");
    if (e->cfg.flags & NL_FLAG_LOCAL_GROUP) return e->fail(NL_ERR_INVALID, "local groups sum in-process; no export");
    if (int rc = p2p_alloc_area(e)) return rc;
    nl_engine::P2P &p = e->p2p;
    hipIpcMemHandle_t h;
    hipError_t s = hipIpcGetMemHandle(&h, p.area);
    if (s != hipSuccess) {
        (void)hipGetLastError();
        return e->fail(NL_ERR_UNSUPPORTED, "push all-reduce: the uncached receive area cannot be exported (hipIpcGetMemHandle: %s); use RCCL", hipGetErrorString(s));
    }
    static_assert(sizeof(hipIpcMemHandle_t) == NL_P2P_HANDLE_BYTES, "handle size");
    memcpy(handle_out, &h, NL_P2P_HANDLE_BYTES);
    return NL_OK;
}

int nl_p2p_import(nl_handle e, const void *handles) {
    if (e && e->grp) return e->fail(NL_ERR_UNSUPPORTED, "nl_p2p_import on a device group: nl_create_group wires its ranks itself");
    if (!e || !handles) return NL_ERR_INVALID;
    if (e->finalized) return e->fail(NL_ERR_STATE, "nl_p2p_import after nl_finalize");
    nl_engine::P2P &p = e->p2p;
    if (!p.area) return e->fail(NL_ERR_STATE, "nl_p2p_import before nl_p2p_export");
    HIPCK(e, hipSetDevice(e->dev));
    for (int r = 0; r < e->G; r++) {
        if (r == e->rank) { p.peer[r] = p.area; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + (size_t)r * NL_P2P_HANDLE_BYTES, NL_P2P_HANDLE_BYTES);
        hipError_t s = hipIpcOpenMemHandle(&p.peer[r], h, hipIpcMemLazyEnablePeerAccess);
        if (s != hipSuccess) return e->fail(NL_ERR_COMM, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(s));
        p.opened[r] = true;
        if (!p.uncached) {   // NL_P2P_CACHED test knob: only sound while every rank shares this device's L2
            hipPointerAttribute_t at{};
            if (hipPointerGetAttributes(&at, p.peer[r]) == hipSuccess && at.device != e->dev)
                return e->fail(NL_ERR_UNSUPPORTED, "NL_P2P_CACHED is a one-device test knob: rank %d lives on device %d", r, at.device);
            (void)hipGetLastError();
        }
    }
    p.on = true;
    return NL_OK;
}

int nl_p2p_loopback(nl_handle e) {
    // Measurement mode: ONE rank of a tensor-parallel group on an otherwise idle GPU.  The launch plan, grids, granule
    // stores and polls are the tensor-parallel ones; every "peer" is this rank's own receive area and the rank writes
    // its partial into all G rank-slots itself, so only the xGMI hop is missing.  The residual stream then holds
    // G x this rank's partial sums: timings are real, logits are NOT a model's.
    if (!e) return NL_ERR_INVALID;
    if (e->grp) return e->fail(NL_ERR_UNSUPPORTED, "nl_p2p_loopback on a device group: a measurement mode of ONE rank");
    if (e->finalized) return e->fail(NL_ERR_STATE, "nl_p2p_loopback after nl_finalize");
    unsigned char h[NL_P2P_HANDLE_BYTES];
    int rc = nl_p2p_export(e, h);            // allocates the (uncached) area exactly as a real rank does
    if (rc) return rc;
    for (int r = 0; r < e->G; r++) e->p2p.peer[r] = e->p2p.area;
    e->p2p.loopback = true;
    e->p2p.on = true;
    return NL_OK;
}

int nl_p2p_info(nl_handle e, int *enabled, int *uncached) {
    if (e && e->grp) return nl_p2p_info(e->grp->members[0], enabled, uncached);
    if (!e) return NL_ERR_INVALID;
    if (enabled) *enabled = e->p2p.on ? 1 : 0;
    if (uncached) *uncached = e->p2p.uncached ? 1 : 0;
    return NL_OK;
}

}  // extern "C"
