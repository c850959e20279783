//go:build hip

// hip_backend.go -- cgo binding of libnanollama_hip.so behind nanollama's Go inference engine.
//
// Drop this file and hip_stub.go into the reference's go/ directory, apply model_hip.patch, and build with
// `go build -tags hip` (after `make -C nanollama_amd/csrc`).  Without the tag the engine is the unmodified
// pure-Go one.  Every entry point used here is declared in include/nanollama_hip.h next to the Go interface it
// replaces.  The build image of this repository has no Go toolchain: the same call sequence is exercised by the
// ctypes host (nanollama_amd/model.py) and by the plain-C driver tests/abi_driver.c.
package main

/*
#cgo CFLAGS: -I${SRCDIR}/../include
#cgo LDFLAGS: -L${SRCDIR}/../nanollama_amd -lnanollama_hip -Wl,-rpath,${SRCDIR}/../nanollama_amd
#include <stdlib.h>
#include "nanollama_hip.h"
*/
import "C"

import (
	"fmt"
	"os"
	"regexp"
	"strconv"
	"unsafe"
)

const hipEnabled = true

// hipGPUs is how many GPUs of the node the model is sharded over: NANOLLAMA_GPUS=2|4|8 (default 1).  One process, one
// LlamaModel, one mutex in go/serve.go -- the sharding is behind the handle (nl_create_group).
func hipGPUs() int {
	n, err := strconv.Atoi(os.Getenv("NANOLLAMA_GPUS"))
	if err != nil || n < 1 {
		return 1
	}
	return n
}

// hipBackend owns the device handle that replaces LlamaWeights and the KV cache of LlamaState.
type hipBackend struct{ h C.nl_handle }

func hipErr(h C.nl_handle, rc C.int, what string) error {
	return fmt.Errorf("%s: %s (nl_status %d)", what, C.GoString(C.nl_last_error(h)), int(rc))
}

// the tensors loadWeights reads (go/model.go:177-265); anything else in the file is ignored, as there
var hipTensor = regexp.MustCompile(`^(token_embd\.weight|output_norm\.weight|output\.weight|blk\.\d+\.((attn_norm|ffn_norm|attn_q|attn_k|attn_v|attn_output|ffn_gate|ffn_up|ffn_down)\.weight|attn_(q|k|v|output)\.bias))$`)

// loadHIP is called from LoadLlamaModel (go/model.go:121) instead of loadWeights / allocState / precomputeRoPE.
// Every tensor is handed over with its raw GGUF bytes; the library copies during the call (cgo pointer rule: no
// Go pointer is retained), re-packs and keeps only what the device needs.
// gpus > 1 shards the model tensor-parallel over devices device .. device+gpus-1 of this process (nl_create_group): the
// handle behaves like a single-device one, so Forward / Reset / the decode fast paths below are unchanged.
func loadHIP(gguf *GGUFFile, cfg *LlamaConfig, device int, gpus int) (*hipBackend, error) {
	c := C.nl_config{
		n_layers: C.int32_t(cfg.NumLayers), dim: C.int32_t(cfg.EmbedDim),
		n_heads: C.int32_t(cfg.NumHeads), n_kv_heads: C.int32_t(cfg.NumKVHeads),
		head_dim: C.int32_t(cfg.HeadDim), interm: C.int32_t(cfg.IntermSize),
		vocab: C.int32_t(cfg.VocabSize), seq_len: C.int32_t(cfg.SeqLen),
		rms_eps: C.float(cfg.RMSNormEps), rope_theta: C.float(cfg.RopeTheta),
		max_streams: 1, device: C.int32_t(device), tp_size: 1,
	}
	if cfg.QKNorm {
		c.qk_norm = 1
	}
	if cfg.RopeConjugate {
		c.rope_conjugate = 1
	}
	b := &hipBackend{}
	if gpus > 1 {
		ids := make([]C.int, gpus)
		for i := range ids {
			ids[i] = C.int(device + i)
		}
		if rc := C.nl_create_group(&c, &ids[0], C.int(gpus), &b.h); rc != 0 {
			return nil, hipErr(nil, rc, "nl_create_group")
		}
	} else if rc := C.nl_create(&c, &b.h); rc != 0 {
		return nil, hipErr(nil, rc, "nl_create")
	}
	for name, info := range gguf.Tensors {
		if !hipTensor.MatchString(name) {
			continue
		}
		data, _, err := gguf.GetTensor(name) // go/gguf.go:561
		if err != nil {
			C.nl_destroy(b.h)
			return nil, err
		}
		rows, cols := C.uint64_t(1), C.uint64_t(info.Dims[0])
		if info.NDims >= 2 { // GGUF dims are innermost-first (go/gguf.go:352-357)
			rows = C.uint64_t(info.Dims[1])
		}
		cname := C.CString(name)
		rc := C.nl_upload_tensor(b.h, cname, C.uint32_t(info.Type), unsafe.Pointer(&data[0]),
			C.uint64_t(len(data)), rows, cols)
		C.free(unsafe.Pointer(cname))
		if rc != 0 {
			err := hipErr(b.h, rc, "nl_upload_tensor "+name)
			C.nl_destroy(b.h)
			return nil, err
		}
	}
	if rc := C.nl_finalize(b.h); rc != 0 {
		err := hipErr(b.h, rc, "nl_finalize")
		C.nl_destroy(b.h)
		return nil, err
	}
	return b, nil
}

// logits is the slice LoadLlamaModel installs as State.Logits (go/model.go:30): the library's pinned host buffer
// (nl_host_logits), so that nl_forward / nl_prefill leave the logits where the sampling code reads and mutates them
// (go/main.go:174-213) without a copy -- C memory, which Go code may read and write freely; it lives until nl_destroy.
func (b *hipBackend) logits(vocab int) []float32 {
	p := C.nl_host_logits(b.h)
	if p == nil {
		return make([]float32, vocab)
	}
	return unsafe.Slice((*float32)(unsafe.Pointer(p)), vocab)
}

// forwardHIP replaces the body of (*LlamaModel).Forward (go/model.go:490-620): State.Logits stays the slice the
// sampling code reads and mutates (go/main.go:174-187).  Consecutive calls are steps of one resident launch for the
// smallest tier (nl_persist_info): no launch per token.
func (m *LlamaModel) forwardHIP(token, pos int) {
	rc := C.nl_forward(m.hip.h, 0, C.int(token), C.int(pos), (*C.float)(unsafe.Pointer(&m.State.Logits[0])))
	if rc != 0 {
		panic(hipErr(m.hip.h, rc, "nl_forward")) // the reference Forward cannot fail; a device error is fatal
	}
}

// resetHIP replaces (*LlamaModel).Reset (go/model.go:623-631); O(1) on the device.
func (m *LlamaModel) resetHIP() {
	C.nl_reset(m.hip.h, 0)
	m.State.Pos = 0
}

// applyGammaHIP hands the gamma essence (go/gamma.go, go/main.go:70-83) to the device: Forward adds
// gamma[token] to the embedding row there (go/model.go:503-505).
func (m *LlamaModel) applyGammaHIP(indices []int32, values []float32) error {
	var ip *C.int32_t
	var vp unsafe.Pointer
	if len(indices) > 0 {
		ip, vp = (*C.int32_t)(unsafe.Pointer(&indices[0])), unsafe.Pointer(&values[0])
	}
	if rc := C.nl_set_gamma(m.hip.h, ip, C.int(len(indices)), vp, 0); rc != 0 {
		return hipErr(m.hip.h, rc, "nl_set_gamma")
	}
	return nil
}

// decodeGreedyHIP is the decode loop of Engine.Generate for temp <= 0 and repPenalty <= 1 (go/main.go:173-219)
// chained on the device: no logits cross the bus, the caller applies the EOS stop by truncating.
func (m *LlamaModel) decodeGreedyHIP(token, pos, n int) []int32 {
	ids := make([]int32, n+1)
	var done C.int
	if rc := C.nl_decode_greedy(m.hip.h, 0, C.int(token), C.int(pos), C.int(n), (*C.int)(unsafe.Pointer(&ids[0])), &done); rc != 0 {
		panic(hipErr(m.hip.h, rc, "nl_decode_greedy"))
	}
	return ids[:int(done)]
}

// prefillHIP is the prompt loop of Engine.Generate (go/main.go:160-166) as one call (matrix-core path for
// Q4_0 / Q8_0 files); State.Logits receives the logits after the last prompt token.
func (m *LlamaModel) prefillHIP(tokens []int32, pos0 int) {
	if len(tokens) == 0 {
		return
	}
	rc := C.nl_prefill(m.hip.h, 0, (*C.int)(unsafe.Pointer(&tokens[0])), C.int(len(tokens)), C.int(pos0),
		(*C.float)(unsafe.Pointer(&m.State.Logits[0])))
	if rc != 0 {
		panic(hipErr(m.hip.h, rc, "nl_prefill"))
	}
}
