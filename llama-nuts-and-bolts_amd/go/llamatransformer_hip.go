//go:build hip

// llamatransformer_hip.go -- package model's LlamaTransformer on the MI355X library (liblnb_hip.so), selected with `-tags hip`.
//
// This file and inferencecontext_hip.go define THE SAME exported names as the reference's src/model/llamatransformer.go and
// src/model/inferencecontext.go that the rest of the reference uses (tests/test_go_binding.py scans every non-test .go file of
// src/model, src/inference and cmd for what it reads from these types and checks that each name is declared here; the files have
// never been through a Go compiler -- there is none in the build image):
//     Model.Transformer *LlamaTransformer                                   src/model/model.go:48
//     NewLlamaTransformer(model *Model) (*LlamaTransformer, error)          src/model/llamatransformer.go:64   (called by the loader)
//     (*LlamaTransformer).Forward(infContext, inputTokens, startPos)        src/model/llamatransformer.go:145  (called at src/inference/inference.go:202)
//     LlamaTransformer.PrecomputedFreqsCis                                  src/model/llamatransformer.go:24
// To adopt: copy both files into src/model/, add the line `//go:build !hip` at the top of the two reference files they replace (and of
// llamatransformer_simulated_test.go, which pokes the CPU implementation's private fields), and build with
//     CGO_CFLAGS="-I<repo>/include" CGO_LDFLAGS="-L<repo>/llama-nuts-and-bolts_amd -llnb_hip -Wl,-rpath,<repo>/llama-nuts-and-bolts_amd" go build -tags hip ./...
// scripts/check_go.sh runs `go vet -tags hip ./src/model/` against a checkout of the reference when a Go toolchain exists.
//
// NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain (`go: command not found`).  The C side of every call
// below is exercised through the same ABI by tests/ (ctypes) and tests/native/host_mirror_test.cpp (C++).

package model

/*
#cgo LDFLAGS: -llnb_hip
#include <stdint.h>
#include <stdlib.h>
#include "lnb.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"sync"
	"unsafe"

	"github.com/adalkiran/llama-nuts-and-bolts/src/ml"
)

// lnb_last_error() is thread-local in the library and a goroutine may migrate between OS threads: the failing call and the fetch of
// its message run with the goroutine pinned to one thread.
func lnbCall(f func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if f() != 0 {
		return errors.New(C.GoString(C.lnb_last_error()))
	}
	return nil
}

// LlamaTransformer keeps the exported surface of the reference struct.  The weights live on the device behind `handle`
// (re-tiled once by lnb_model_set_tensor); Layers stays exported with one entry per block for code that ranges over it.
type LlamaTransformer struct {
	Layers []*LlamaTransformerBlock

	PrecomputedFreqsCis *ml.Tensor // [2*MaxSequenceLength, HeadDim/2] complex64, as the reference computes it (read back from the library)

	handle *C.lnb_model
	args   *ModelArgs
	mu     sync.Mutex
	ctxs   int // live InferenceContexts (Close refuses while > 0)
}

// LlamaTransformerBlock: the reference's blocks hold weight tensors; here the weights are on the device and a block keeps what the
// rest of package model reads from it: its index and the two derived dimensions printModelInfo prints through the private
// attention / feedForward fields (src/model/loader.go:156-163; field names as in src/model/llamatransformer.go:27-57).
type LlamaTransformerBlock struct {
	LayerIndex int

	attention   *LlamaAttention
	feedForward *LlamaFeedForward
}

// LlamaAttention: the head geometry of src/model/llamatransformer.go:37-43 (no weight tensors on the host).
type LlamaAttention struct {
	LayerIndex int

	N_Heads   int
	N_KVHeads int
	N_Rep     int
	HeadDim   int
}

// LlamaFeedForward: FFNHiddenDim as derived by src/model/llamatransformer.go:569-577 (here: lnb_model_ffn_hidden_dim).
type LlamaFeedForward struct {
	FFNHiddenDim int
}

func boolToC(b bool) C.int32_t {
	if b {
		return 1
	}
	return 0
}

// NewLlamaTransformer binds every checkpoint tensor by its Meta key (the names and shapes getTensor / getLayerTensor check,
// src/model/loader.go:183-192) and builds the RoPE table; same signature and error behaviour as src/model/llamatransformer.go:64-113.
func NewLlamaTransformer(model *Model) (*LlamaTransformer, error) {
	// the binding was generated against include/lnb.h's LNB_ABI_VERSION: a stale liblnb_hip.so on the library path must fail here, not write
	// through a mistyped pointer later
	if v := int(C.lnb_abi_version()); v != int(C.LNB_ABI_VERSION) {
		return nil, fmt.Errorf("liblnb_hip.so reports ABI version %d, this binding was built against %d", v, int(C.LNB_ABI_VERSION))
	}
	a := model.ModelArgs
	cargs := C.lnb_model_args{
		dim: C.int32_t(a.Dim), n_layers: C.int32_t(a.N_Layers), n_heads: C.int32_t(a.N_Heads), n_kv_heads: C.int32_t(a.N_KVHeads),
		vocab_size: C.int32_t(a.VocabSize), multiple_of: C.int32_t(a.MultipleOf), ffn_dim_multiplier: C.double(a.FFNDimMultiplier),
		norm_eps: C.float(a.NormEpsilon), use_scaled_rope: boolToC(a.UseScaledRope), rope_theta: C.double(a.RopeTheta),
		max_seq_len: C.int32_t(a.MaxSequenceLength),
	}
	lt := &LlamaTransformer{args: a}
	if err := lnbCall(func() C.int { return C.lnb_model_create(&cargs, 0, 0, C.int(a.N_Layers), &lt.handle) }); err != nil {
		return nil, err
	}
	n := int(C.lnb_model_num_tensors(lt.handle))
	for k := 0; k < n; k++ {
		var cname *C.char
		var shape [2]C.int64_t
		var rank C.int
		if err := lnbCall(func() C.int { return C.lnb_model_tensor_info(lt.handle, C.int(k), &cname, &shape[0], &rank) }); err != nil {
			lt.Close()
			return nil, err
		}
		name := C.GoString(cname)
		tensor, ok := model.Tensors.Get(name)
		if !ok {
			lt.Close()
			return nil, fmt.Errorf("tensor \"%s\" not found", name) // loader.go:185
		}
		tshape := make([]C.int64_t, len(tensor.Size))
		for i, s := range tensor.Size {
			tshape[i] = C.int64_t(s)
		}
		// RawData is a sub-slice of the checkpoint mmap (src/torch/types.go:51-55): copied to the device, not retained
		if err := lnbCall(func() C.int {
			return C.lnb_model_set_tensor(lt.handle, cname, (*C.uint16_t)(unsafe.Pointer(&tensor.RawData[0])), &tshape[0], C.int(len(tshape)))
		}); err != nil {
			lt.Close()
			return nil, err
		}
	}
	if err := lnbCall(func() C.int { return C.lnb_model_finalize(lt.handle, 0) }); err != nil {
		lt.Close()
		return nil, err
	}
	// PrecomputedFreqsCis (exported field, llamatransformer.go:24,109): [rows, HeadDim/2] complex64 = f32 pairs
	var rows C.int
	if err := lnbCall(func() C.int { return C.lnb_model_rope_table(lt.handle, nil, 0, &rows) }); err != nil {
		lt.Close()
		return nil, err
	}
	lt.PrecomputedFreqsCis = ml.NewEmptyTensor([]int{int(rows), a.HeadDim / 2}, ml.DT_COMPLEX)
	nf := C.int64_t(int(rows) * (a.HeadDim / 2) * 2)
	if err := lnbCall(func() C.int {
		return C.lnb_model_rope_table(lt.handle, (*C.float)(unsafe.Pointer(&lt.PrecomputedFreqsCis.RawData[0])), nf, &rows)
	}); err != nil {
		lt.Close()
		return nil, err
	}
	nKVHeads := a.N_KVHeads
	if nKVHeads < 0 { // llamatransformer.go:73-75
		nKVHeads = a.N_Heads
	}
	ffnHiddenDim := int(C.lnb_model_ffn_hidden_dim(&cargs))
	lt.Layers = make([]*LlamaTransformerBlock, a.N_Layers)
	for i := range lt.Layers {
		lt.Layers[i] = &LlamaTransformerBlock{
			LayerIndex:  i,
			attention:   &LlamaAttention{LayerIndex: i, N_Heads: a.N_Heads, N_KVHeads: nKVHeads, N_Rep: a.N_Heads / nKVHeads, HeadDim: a.Dim / a.N_Heads},
			feedForward: &LlamaFeedForward{FFNHiddenDim: ffnHiddenDim},
		}
	}
	runtime.SetFinalizer(lt, func(t *LlamaTransformer) { t.Close() })
	return lt, nil
}

// Close frees the device copy.  It refuses while InferenceContexts created on this transformer are alive (they hold device
// buffers that refer to it); contexts keep a Go reference to the transformer, so the garbage collector finalizes them first.
func (lt *LlamaTransformer) Close() error {
	lt.mu.Lock()
	defer lt.mu.Unlock()
	if lt.handle == nil {
		return nil
	}
	if lt.ctxs > 0 {
		return fmt.Errorf("LlamaTransformer.Close: %d InferenceContext(s) still open", lt.ctxs)
	}
	C.lnb_model_destroy(lt.handle)
	lt.handle = nil
	return nil
}

// Forward: same contract as src/model/llamatransformer.go:145-180 -- inputTokens [seq] DT_INT32, logits [seq, VocabSize] DT_F32
// (bf16-representable values), errors for an empty token array and for positions beyond the RoPE table / KV cache.
func (lt *LlamaTransformer) Forward(infContext *InferenceContext, inputTokens *ml.Tensor, startPos int) (*ml.Tensor, error) {
	if inputTokens.Size[0] == 0 {
		return nil, fmt.Errorf("empty token array") // llamatransformer.go:146-148
	}
	if err := infContext.attach(lt); err != nil {
		return nil, err
	}
	seq := inputTokens.Size[0]
	output := ml.NewEmptyTensor([]int{seq, lt.args.VocabSize}, ml.DT_F32)
	release, err := infContext.installLayerHook() // per-layer Logf (llamatransformer.go:163), only while this call runs
	if err != nil {
		return nil, err
	}
	defer release()
	if err := lnbCall(func() C.int {
		return C.lnb_forward(infContext.handle, (*C.int32_t)(unsafe.Pointer(&inputTokens.RawData[0])), C.int(seq), C.int(startPos),
			(*C.float)(unsafe.Pointer(&output.RawData[0])), nil)
	}); err != nil {
		return nil, err
	}
	if infContext.MirrorCaches {
		if err := infContext.SyncCachesFromDevice(); err != nil {
			return nil, err
		}
	}
	runtime.KeepAlive(inputTokens)
	return output, nil
}
