// llamatransformer_hip.go -- cgo binding of the MI355X-native LlamaTransformer.Forward path (liblnb_hip.so).
//
// UNTESTED SOURCE: no Go toolchain exists in the build image (`go: command not found`), so this file has never been
// compiled.  It shows the exact binding a maintainer of adalkiran/llama-nuts-and-bolts adds to package `model`
// (src/model/) to swap the CPU path for the HIP library while keeping every exported Go signature:
//
//   NewLlamaTransformer(model *Model) (*LlamaTransformer, error)                       src/model/llamatransformer.go:64
//   (*LlamaTransformer).Forward(infContext, inputTokens *ml.Tensor, startPos int)       src/model/llamatransformer.go:145
//   NewInferenceContext(model, inferenceArgs, logFn) *InferenceContext                  src/model/inferencecontext.go:17
//
// Build:  CGO_CFLAGS="-I<repo>/include" CGO_LDFLAGS="-L<repo>/llama-nuts-and-bolts_amd -llnb_hip" go build -tags hip ./...
//go:build hip

package model

/*
#cgo LDFLAGS: -llnb_hip
#include <stdlib.h>
#include "lnb.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"

	"github.com/adalkiran/llama-nuts-and-bolts/src/common"
	"github.com/adalkiran/llama-nuts-and-bolts/src/ml"
)

func lastError() error { return errors.New(C.GoString(C.lnb_last_error())) }

// LlamaTransformerHIP replaces the weight-holding fields of LlamaTransformer; the exported fields the tests read
// (Layers, PrecomputedFreqsCis) are kept on the embedding struct.
type LlamaTransformerHIP struct {
	handle *C.lnb_model
	args   *ModelArgs
}

func newLlamaTransformerHIP(model *Model, device int) (*LlamaTransformerHIP, error) {
	a := model.ModelArgs
	cargs := C.lnb_model_args{
		dim: C.int32_t(a.Dim), n_layers: C.int32_t(a.N_Layers), n_heads: C.int32_t(a.N_Heads), n_kv_heads: C.int32_t(a.N_KVHeads),
		vocab_size: C.int32_t(a.VocabSize), multiple_of: C.int32_t(a.MultipleOf), ffn_dim_multiplier: C.double(a.FFNDimMultiplier),
		norm_eps: C.float(a.NormEpsilon), use_scaled_rope: boolToC(a.UseScaledRope), rope_theta: C.double(a.RopeTheta),
		max_seq_len: C.int32_t(a.MaxSequenceLength),
	}
	t := &LlamaTransformerHIP{args: a}
	if C.lnb_model_create(&cargs, C.int(device), 0, C.int(a.N_Layers), &t.handle) != 0 {
		return nil, lastError()
	}
	// bind every checkpoint tensor by its Meta key: same names and shapes getTensor/getLayerTensor check (loader.go:183-192).
	// RawData is a sub-slice of the mmap (src/torch/types.go:51-55): the library copies it to the device and keeps nothing.
	for _, name := range model.Tensors.GetKeys() {
		tensor, _ := model.Tensors.Get(name)
		shape := make([]C.int64_t, len(tensor.Size))
		for i, s := range tensor.Size {
			shape[i] = C.int64_t(s)
		}
		cname := C.CString(name)
		rc := C.lnb_model_set_tensor(t.handle, cname, (*C.uint16_t)(unsafe.Pointer(&tensor.RawData[0])), &shape[0], C.int(len(shape)))
		C.free(unsafe.Pointer(cname))
		if rc != 0 {
			C.lnb_model_destroy(t.handle)
			return nil, lastError()
		}
	}
	if C.lnb_model_finalize(t.handle, 0) != 0 { // PrecomputedFreqsCis, llamatransformer.go:109
		C.lnb_model_destroy(t.handle)
		return nil, lastError()
	}
	runtime.SetFinalizer(t, func(t *LlamaTransformerHIP) { C.lnb_model_destroy(t.handle) })
	return t, nil
}

// InferenceContextHIP is the device KV cache behind model.InferenceContext (inferencecontext.go:8-15).
type InferenceContextHIP struct {
	handle         *C.lnb_ctx
	SequenceLength int
}

func newInferenceContextHIP(t *LlamaTransformerHIP, inferenceArgs common.InferenceArgs) (*InferenceContextHIP, error) {
	c := &InferenceContextHIP{SequenceLength: inferenceArgs.SequenceLength}
	if C.lnb_ctx_create(t.handle, C.int(inferenceArgs.SequenceLength), &c.handle) != 0 {
		return nil, lastError()
	}
	runtime.SetFinalizer(c, func(c *InferenceContextHIP) { C.lnb_ctx_destroy(c.handle) })
	return c, nil
}

// Forward is the body of (*LlamaTransformer).Forward (llamatransformer.go:145-180) on the HIP path:
// inputTokens is the DT_INT32 tensor the generation loop slices (inference.go:195), the result is the
// [sequenceLength, VocabSize] DT_F32 logits tensor the caller argmaxes (inference.go:207-211).
func (t *LlamaTransformerHIP) Forward(infContext *InferenceContextHIP, inputTokens *ml.Tensor, startPos int) (*ml.Tensor, error) {
	if inputTokens.Size[0] == 0 {
		return nil, fmt.Errorf("empty token array")
	}
	if inputTokens.DataType != ml.DT_INT32 {
		return nil, fmt.Errorf("tensor is not in data type %s: \"%s\" is %s", ml.DT_INT32, inputTokens.Name, inputTokens.DataType)
	}
	seq := inputTokens.Size[0]
	logits := ml.NewEmptyTensor([]int{seq, t.args.VocabSize}, ml.DT_F32)
	var pinner runtime.Pinner // Go memory handed to C for the duration of the call
	pinner.Pin(&inputTokens.RawData[0])
	pinner.Pin(&logits.RawData[0])
	defer pinner.Unpin()
	rc := C.lnb_forward(infContext.handle, (*C.int32_t)(unsafe.Pointer(&inputTokens.RawData[0])), C.int(seq), C.int(startPos),
		(*C.float)(unsafe.Pointer(&logits.RawData[0])), nil)
	if rc != 0 {
		return nil, lastError()
	}
	return logits, nil
}

// DecodeGreedy runs n one-token Forward+Argmax steps on the device (the loop body of inference.go:194-252).
func (t *LlamaTransformerHIP) DecodeGreedy(infContext *InferenceContextHIP, token TokenId, startPos int, n int) ([]TokenId, error) {
	out := make([]int32, n)
	if C.lnb_decode_greedy(infContext.handle, C.int32_t(token), C.int(startPos), C.int(n), (*C.int32_t)(unsafe.Pointer(&out[0])), nil) != 0 {
		return nil, lastError()
	}
	res := make([]TokenId, n)
	for i, v := range out {
		res[i] = TokenId(v)
	}
	return res, nil
}

func boolToC(b bool) C.int32_t {
	if b {
		return 1
	}
	return 0
}
