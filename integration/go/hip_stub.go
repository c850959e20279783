//go:build !hip

// hip_stub.go -- the pure-Go build: no device backend, the hooks of model_hip.patch are dead code.
package main

const hipEnabled = false

type hipBackend struct{}

func hipGPUs() int                                                              { return 1 }
func loadHIP(gguf *GGUFFile, cfg *LlamaConfig, device int, gpus int) (*hipBackend, error) {
	return nil, nil
}
func (b *hipBackend) logits(vocab int) []float32                                { return nil }
func (m *LlamaModel) forwardHIP(token, pos int)                                 {}
func (m *LlamaModel) resetHIP()                                                 {}
