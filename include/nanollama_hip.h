/*
 * nanollama_hip.h -- C ABI of libnanollama_hip.so, the MI355X (gfx950) engine
 * behind the nanollama Go inference surface.
 *
 * The reference (ariannamethod/nanollama, go/) is single-package pure Go with
 * no FFI; the seams this ABI replaces are the three methods the Go host calls
 * on its model object:
 *
 *   LoadLlamaModel(gguf)            go/model.go:121   -> nl_create + nl_upload_tensor* + nl_finalize
 *   (*LlamaModel).Forward(tok,pos)  go/model.go:490   -> nl_forward   (fills the caller's Logits slice)
 *   (*LlamaModel).Reset()           go/model.go:623   -> nl_reset
 *   argmax(State.Logits)            go/main.go:400    -> nl_forward_argmax / nl_decode_greedy (greedy fast path)
 *
 * Conventions: every function returns 0 on success or a negative nl_status;
 * nl_last_error(h) gives the message.  No exceptions cross the boundary, no
 * caller pointer is retained after a call returns (cgo pointer rule), all
 * sizes are explicit.  A handle is single-caller / non-reentrant, like the Go
 * engine (one Engine per process, HTTP handler under a mutex, go/serve.go:56).
 * Placement: nl_create drives ONE GPU; a model sharded tensor-parallel over the
 * GPUs of a node is either ONE process holding a group handle (nl_create_group:
 * what the drop-in Go host uses, because the reference is one process) or one
 * process per GPU (tp_rank / tp_size) joined by the push all-reduce
 * (nl_p2p_export / nl_p2p_import) or an RCCL communicator (nl_comm_*).
 */
#ifndef NANOLLAMA_HIP_H
#define NANOLLAMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NL_API __attribute__((visibility("default")))

typedef struct nl_engine *nl_handle;

typedef enum {
    NL_OK = 0,
    NL_ERR_INVALID = -1,      /* bad argument / shape */
    NL_ERR_UNSUPPORTED = -2,  /* tensor type or feature not implemented on device */
    NL_ERR_HIP = -3,          /* HIP runtime error (message has the hipError string) */
    NL_ERR_STATE = -4,        /* call order violated (e.g. forward before finalize) */
    NL_ERR_MISSING = -5,      /* a required tensor was never uploaded */
    NL_ERR_COMM = -6          /* RCCL failure */
} nl_status;

/* ggml tensor types accepted by nl_upload_tensor (go/gguf.go:43-57). */
enum { NL_GGML_F32 = 0, NL_GGML_F16 = 1, NL_GGML_Q4_0 = 2, NL_GGML_Q5_0 = 6, NL_GGML_Q8_0 = 8, NL_GGML_Q4_K = 12,
       NL_GGML_Q6_K = 14 };  /* every format go/quant.go multiplies with */

/* Mirrors LlamaConfig (go/model.go:27-42) + placement.  head_dim == 0 means
 * dim / n_heads (go/model.go:140-142); seq_len is capped to 2048 exactly as
 * go/model.go:145-148 does. */
typedef struct {
    int32_t n_layers, dim, n_heads, n_kv_heads, head_dim, interm, vocab, seq_len;
    float rms_eps, rope_theta;
    int32_t qk_norm, rope_conjugate;
    int32_t max_streams; /* independent KV caches (decode streams); 0 -> 1 */
    int32_t device;      /* HIP device ordinal of this process */
    int32_t tp_rank, tp_size; /* tensor-parallel shard; tp_size 0/1 -> single GPU */
    int32_t flags;       /* NL_FLAG_* */
} nl_config;

#define NL_FLAG_NO_GRAPH 1 /* launch kernels eagerly instead of replaying a hipGraph */
#define NL_FLAG_LOCAL_GROUP 2 /* tp_size > 1 shards living in ONE process, stepped by nl_group_forward */
#define NL_FLAG_GROUP_FUSED 4 /* with NL_FLAG_LOCAL_GROUP: short contexts step the two-launches-per-layer plan of a push-group rank */

/* Library / device probes (no handle, no device work for nl_abi_version). */
NL_API int nl_abi_version(void);
NL_API int nl_device_count(void);
/* "src=<sha256[:16] of the library's sources at build time> git=<HEAD at build time>" */
NL_API const char *nl_build_info(void);

/* == LoadLlamaModel (go/model.go:121-174) ================================== */
NL_API int nl_create(const nl_config *cfg, nl_handle *out);
/* LoadLlamaModel for a model sharded tensor-parallel over n GPUs of THIS process (go/main.go:63 LoadLlamaModel is one process,
 * go/serve.go:54-108 one engine behind a mutex; BASELINE's 7.9B tier shards across the node's 8 GPUs).  The handle is used
 * exactly like nl_create's: nl_upload_tensor takes the FULL tensors (every rank keeps its rows / columns), then nl_finalize,
 * nl_forward, nl_forward_argmax, nl_decode_greedy, nl_sample_decode, nl_prefill, nl_forward_batch, nl_reset, nl_destroy.
 * Behind it: n rank engines (tp_rank r on device_ids[r]; n = 2, 4 or 8), heads / FFN rows / vocabulary split by rows,
 * WO / down by columns, the 2 L seams of a token closed by the push all-reduce (above nl_p2p_export) through plain peer
 * pointers (hipDeviceEnablePeerAccess), each rank stepped by its own host thread.  cfg->tp_size is ignored (or must equal n),
 * cfg->device is ignored.  device_ids may repeat -- every rank on one device is the one-GPU test configuration, bitwise
 * equal to the in-process shard group below.  Errors: NL_ERR_UNSUPPORTED when a device cannot reach a peer. */
NL_API int nl_create_group(const nl_config *cfg, const int *device_ids, int n, nl_handle *out);
/* loadWeights (go/model.go:177-265): one call per GGUF tensor, by its GGUF
 * name, with the FULL tensor as stored in the file (raw block bytes, rows x
 * cols in row-major [out,in] order; 1-D tensors: rows = 1).  The bytes are
 * copied during the call; the library re-packs them for the device and keeps
 * only this rank's tensor-parallel shard.  Unknown names -> NL_ERR_INVALID,
 * unsupported types -> NL_ERR_UNSUPPORTED (the Go engine would print a WARNING
 * and compute garbage, go/model.go:383-385; we refuse instead).  Optional
 * blk.N.attn_{q,k,v,output}.bias tensors (go/model.go:244-247) are accepted as F32/F16, each independently
 * optional (an absent q/k/v bias is a zero vector).  Callers skip tensors the Go loader never reads. */
NL_API int nl_upload_tensor(nl_handle h, const char *gguf_name, uint32_t ggml_type, const void *data,
                            uint64_t nbytes, uint64_t rows, uint64_t cols);
/* allocState + precomputeRoPE (go/model.go:324-358), tied-embedding fallback
 * (:195-201), graph capture.  NL_ERR_MISSING if a required tensor is absent. */
NL_API int nl_finalize(nl_handle h);
NL_API int nl_destroy(nl_handle h);
/* Gamma essence (go/gamma.go:22-35, ApplyToEmbedding :272-290; go/main.go:70-83): embed[token] += gamma[token]
 * for the n listed token ids.  values: [n][dim] float32, or raw IEEE binary16 when is_f16.  May be called before or
 * after nl_finalize; n == 0 removes it. */
NL_API int nl_set_gamma(nl_handle h, const int32_t *indices, int n, const void *values, int is_f16);
NL_API const char *nl_last_error(nl_handle h); /* h may be NULL: last create error */

/* == Reset (go/model.go:623-631) ========================================== */
/* O(1): instead of zeroing 2*L*S*kvDim floats the library keeps a per-stream high-water mark (positions
 * [0, mark) written since the last reset) and drops it here.  A later step at pos > mark is legal, as in the
 * reference, and sees what the reference sees: the rows [mark, pos) are cleared to zero just before that step. */
NL_API int nl_reset(nl_handle h, int stream);

/* == Forward (go/model.go:490-620) ======================================== */
/* One token through all layers at position pos of KV stream `stream`;
 * logits_out receives `vocab` floats (caller-owned, e.g. the Go State.Logits
 * slice).  Synchronous on return. */
NL_API int nl_forward(nl_handle h, int stream, int token, int pos, float *logits_out);
/* The library's own pinned host buffer of `vocab` floats (valid from nl_finalize to nl_destroy).  Passed as logits_out,
 * nl_forward / nl_prefill leave the logits there without the copy into a caller-owned slice: the LM head's compute units
 * store them into it directly.  The cgo shim points State.Logits (go/model.go:30, read by go/main.go:186-213) at it.
 * NULL before nl_finalize. */
NL_API float *nl_host_logits(nl_handle h);
/* Forward + argmax (strict '>' => lowest index wins ties, go/main.go:400-408)
 * without moving the logits to the host. */
NL_API int nl_forward_argmax(nl_handle h, int stream, int token, int pos, int *next_id);
/* The greedy decode loop of Engine.Generate (go/main.go:173-219 with temp 0,
 * rep-penalty 1.0) chained on the device: feeds `token` at `pos`, then each
 * sampled id back in, n_steps times, with no host round trip in between.
 * ids_out[i] is the id sampled after step i.  Stops early (returning the
 * count in *n_done) when pos reaches seq_len, as go/main.go:216 does. */
NL_API int nl_decode_greedy(nl_handle h, int stream, int token, int pos, int n_steps, int *ids_out, int *n_done);

/* The sampling half of Engine.Generate's decode loop (go/main.go:173-219) on the device, for temp > 0 and / or
 * rep_penalty > 1: starting from the logits the last nl_forward / nl_prefill of `stream` left on the device
 * (those of position pos - 1), repeat n_steps times { repetition penalty over the recent window (:177-187);
 * sampleTopP when top_p < 1, else sampleTopK (:191-195, :294-398; temp <= 0 -> argmax); append to the window
 * (:197-200); Forward(sampled, pos++) (:213) } with no host round trip and no read-back of the V logits.
 * uniforms[i] (float32 in [0,1)) is the e.rng.Float32() of step i -- the host owns the generator.
 * recent / n_recent (in/out, capacity rep_window <= 1024) is recentTokens.  ids_out[i] is the id sampled at
 * step i; the caller applies the EOS stop (:203) by truncating (steps after an EOS are wasted work, so call in
 * chunks).  Stops early, *n_done < n_steps, when pos reaches seq_len (:216). */
typedef struct {
    float temperature, top_p;
    int32_t top_k;
    float rep_penalty;
    int32_t rep_window;
} nl_sample_params;
NL_API int nl_sample_decode(nl_handle h, int stream, int pos, int n_steps, const nl_sample_params *p,
                            const float *uniforms, int *recent, int *n_recent, int *ids_out, int *n_done);

/* Prompt prefill (go/main.go:160-166 feeds the prompt token-at-a-time through Forward): runs
 * tokens[0..n) at positions pos0..pos0+n-1 of `stream` back to back on the device with no host
 * round trip; last_logits_out (may be NULL) receives the logits after the last token.  Q4_0 / Q8_0 / Q5_0 / F16
 * models take the multi-token matrix-core path (up to 2048 tokens per step): the KV
 * cache and logits agree with n nl_forward calls to the stated logit tolerance (1e-4, summation order
 * and an fp16 hi+lo activation split differ), greedy continuations are identical; other formats run
 * the single-token plan n times (bitwise equal).  pos0 + n must be <= seq_len. */
NL_API int nl_prefill(nl_handle h, int stream, const int *tokens, int n, int pos0, float *last_logits_out);
/* One Forward for each of n (stream, token, pos) triples -- the concurrent decode streams of a
 * serving host.  logits_out (may be NULL) is n x vocab; next_ids (may be NULL) receives each
 * stream's greedy argmax.  Streams must be distinct. */
NL_API int nl_forward_batch(nl_handle h, const int *streams, const int *tokens, const int *pos, int n,
                            float *logits_out, int *next_ids);

/* == introspection / measurement ========================================== */
NL_API int nl_get_config(nl_handle h, nl_config *out); /* effective config (after seq_len cap etc.) */
NL_API int nl_synchronize(nl_handle h);
/* HIP events recorded on the engine's own stream (torch.cuda.Event would not
 * see it): start/stop bracket any sequence of calls; elapsed in milliseconds. */
NL_API int nl_timer_start(nl_handle h);
NL_API int nl_timer_stop(nl_handle h, float *ms);
/* Per-kernel-kind device time of one eager Forward (events around every
 * launch, each replayed `iters` times).  kinds: see nl_kernel_kind_name.  ms_out/calls_out have NL_NUM_KINDS
 * entries.  Measurement only: the replays leave x, the logits and the K/V rows at `pos` in a state no Forward
 * produces, so the stream's positions >= pos count as unwritten afterwards (re-run them before decoding on). */
#define NL_NUM_KINDS 11
NL_API const char *nl_kernel_kind_name(int kind);
NL_API int nl_profile_forward(nl_handle h, int stream, int token, int pos, int iters, float *ms_out, int *calls_out);
/* Which launch plans the handle holds (no reference counterpart; tests and bench.py report it).  fused_mode: 0 = only the
 * general five-launches-per-layer plan; 1 = small tiers, attention block + feed-forward block (2 per layer); 2 = wide
 * tiers, projection + attention fused (4 per layer); 3 = a tensor-parallel rank's layer as two launches with the push
 * all-reduce finished in their tails; 4 = wide tiers on one GPU, projection + attention + WO fused (3 per layer).  fused_max_pos: steps below this position take the fused plan.  launches_*: kernel
 * launches per token of the fused / general plan (0 when the plan does not exist). */
NL_API int nl_plan_info(nl_handle h, int *fused_mode, int *fused_max_pos, int *launches_fused, int *launches_general);
/* The weight-stationary persistent decode of the smallest tier (nl_persist.h; no reference counterpart, the Go loop is
 * go/main.go:173-219): ready = nl_decode_greedy takes ONE launch per chunk while the chunk ends below max_pos (Q8_0 files of
 * nano's shape class on a 256-CU device), and consecutive nl_forward / nl_forward_argmax calls below max_pos are steps of
 * ONE resident launch that takes each call's token from a pinned mailbox word (it leaves when another entry point is called,
 * or NL_PERSIST_IDLE_US -- 2000 by default -- after its last token); launches / tokens count what went through it. */
NL_API int nl_persist_info(nl_handle h, int *ready, int *max_pos, long long *launches, long long *tokens);
/* Device bytes held by the handle (weights, KV, state). */
NL_API int nl_memory_usage(nl_handle h, uint64_t *weights, uint64_t *kv_cache, uint64_t *state);
/* Read back a device state buffer for tests ("x","q","xb2","hb","logits",
 * "k_cache","v_cache"); returns floats copied (<= max_floats). */
NL_API int64_t nl_debug_read(nl_handle h, const char *which, int stream, float *out, int64_t max_floats);

/* Shader-clock phase stamps of one GEMV launch (developer tool, tools/phase_probe.py). */
NL_API int nl_debug_stamps(nl_handle h, int kind, long long *out /* 128 */);

/* == op-level entry points (parity tests of single kernels) =============== */
/* matmulDispatch (go/model.go:361-386): out[rows] = W[rows,cols] @ x[cols],
 * W given as raw GGUF block bytes; x/out are host pointers.  Uses the same
 * re-pack + GEMV kernel as Forward. */
NL_API int nl_op_matmul(int device, uint32_t ggml_type, const void *w, uint64_t nbytes, const float *x, float *out,
                        int rows, int cols);
/* The same product for n_tokens input vectors at once through the MFMA path (x: [n_tokens][cols],
 * out: [n_tokens][rows], host pointers); Q4_0 / Q8_0 / Q5_0 / F16. */
NL_API int nl_op_matmul_batch(int device, uint32_t ggml_type, const void *w, uint64_t nbytes, const float *x,
                               float *out, int rows, int cols, int n_tokens);
/* RMSNormInto (go/quant.go:597-607). */
NL_API int nl_op_rmsnorm(int device, const float *x, const float *w, float eps, float *out, int n);
/* float32(math.Exp(float64(x))) of Softmax and SiLU (go/quant.go:619, :629-631) as the forward kernels compute it
 * (short-chain float64 exponential, nl_kernels.h exp_f64_as_f32), element-wise on host arrays. */
NL_API int nl_op_exp(int device, const float *x, float *out, int n);
/* Soak of the 16-byte {tag, v0, v1, v2} granules the fused launches exchange (nl_tp.h; no reference counterpart): n writer
 * blocks store `iters` generations each, n reader blocks on other XCDs count every copy they read (*seen) and every copy whose
 * four words are not of one generation (*torn; the engine assumes 0). */
NL_API int nl_op_gran16_soak(int device, int n, unsigned iters, unsigned long long *torn, unsigned long long *seen);

/* Sampling operator on host logits: one decision of sampleTopP / sampleTopK / argmax (go/main.go:294-408)
 * after the in-place repetition penalty (:177-187), on the device.  logits (in/out: the penalty is applied in
 * place, as in the reference) is [vocab]; recent (in/out, capacity rep_window) / n_recent follow :197-200. */
NL_API int nl_op_sample(int device, float *logits, int vocab, const nl_sample_params *p, float uniform,
                        int *recent, int *n_recent, int *picked);

/* == tensor-parallel communicator (RCCL over xGMI) ========================= */
#define NL_COMM_ID_BYTES 128
/* Rank 0 obtains an id, the host distributes it (any side channel), every
 * rank passes it to nl_comm_init before nl_finalize. */
NL_API int nl_comm_get_unique_id(void *id_out /* NL_COMM_ID_BYTES */);
NL_API int nl_comm_init(nl_handle h, const void *id /* NL_COMM_ID_BYTES */);

/* Push all-reduce over xGMI (preferred data path of a tensor-parallel run; RCCL above stays as the fallback).
 * The two seams per layer where go/model.go:590-594 and :609-612 add a projection into the residual stream carry a
 * partial [dim] vector per rank.  Instead of 2*L latency-bound 16 KiB ring all-reduces, every rank writes its partial
 * straight into a receive slot inside every peer's memory (one xGMI hop, 8-byte self-validating {tag, value} words)
 * and sums the tp_size slots itself in rank order; the greedy argmax and the logits all-gather travel the same way.
 * Setup, before nl_finalize, one process per GPU: every rank calls nl_p2p_export (allocates its receive area, returns a
 * hipIpc handle), the host all-gathers the NL_P2P_HANDLE_BYTES handles over any side channel, every rank passes the
 * rank-ordered array to nl_p2p_import.  tp_size must be 2, 4 or 8.  Every wait on a peer is bounded (NL_P2P_TIMEOUT_MS,
 * default 10 s); a rank that gives up makes its calls return NL_ERR_COMM. */
#define NL_P2P_HANDLE_BYTES 64
NL_API int nl_p2p_export(nl_handle h, void *handle_out /* NL_P2P_HANDLE_BYTES */);
NL_API int nl_p2p_import(nl_handle h, const void *handles /* tp_size x NL_P2P_HANDLE_BYTES, rank order */);
NL_API int nl_p2p_info(nl_handle h, int *enabled, int *uncached_area);
/* Measurement only (bench.py --shard-of N): this rank alone on an idle GPU with the tensor-parallel launch plan of a
 * tp_size-rank group; the rank writes its partial into all tp_size receive slots of its own area, so grids, granule
 * stores and polls are the real ones and only the xGMI hop is missing.  Timings are real, logits are not a model's.
 * Before nl_finalize, instead of nl_p2p_export / nl_p2p_import.  No reference counterpart (the Go engine is one process). */
NL_API int nl_p2p_loopback(nl_handle h);

/* In-process tensor-parallel group: `n` handles created with tp_size = n, tp_rank = 0..n-1 and
 * NL_FLAG_LOCAL_GROUP (all on devices this process can reach; the same device is allowed).  Steps
 * every shard's launch plan in lockstep and performs the all-reduce / all-gather seams itself, so the
 * sharding arithmetic can be verified on a single GPU.  Same result contract as nl_forward. */
NL_API int nl_group_forward(nl_handle *shards, int n, int stream, int token, int pos, float *logits_out);

#ifdef __cplusplus
}
#endif
#endif /* NANOLLAMA_HIP_H */
