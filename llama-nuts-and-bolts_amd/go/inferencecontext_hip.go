//go:build hip

// inferencecontext_hip.go -- package model's InferenceContext on the MI355X library, selected with `-tags hip` (see
// llamatransformer_hip.go for how the two files slot under the reference's types).  Same exported names as
// src/model/inferencecontext.go:8-52: SequenceLength, CacheK, CacheV, NewInferenceContext(model, inferenceArgs, logFn), Logf.
//
// The KV cache lives on the device.  CacheK / CacheV are HOST MIRRORS with the reference's shape [SequenceLength, N_KVHeads, HeadDim]
// DT_BF16, allocated zero-filled like the reference (inferencecontext.go:32-42) and refreshed from the device by
// SyncCachesFromDevice -- after every Forward when MirrorCaches is set (what a test that reads CacheK, as
// llamatransformer_simulated_test.go:534-538 does, switches on), never otherwise (the copy is 2 x 64 KiB per token and layer).
//
// NOT COMPILED IN THIS REPOSITORY'S CI (no Go toolchain in the build image).

package model

/*
#include <stdint.h>
#include "lnb.h"
extern void lnbGoLayerCallback(int layer, int nLayers, double secs, void* user);
*/
import "C"

import (
	"fmt"
	"runtime"
	"runtime/cgo"
	"unsafe"

	"github.com/adalkiran/llama-nuts-and-bolts/src/common"
	"github.com/adalkiran/llama-nuts-and-bolts/src/ml"
)

type InferenceContext struct {
	SequenceLength int // context size used during inference

	CacheK []*ml.Tensor // host mirrors, see SyncCachesFromDevice
	CacheV []*ml.Tensor

	MirrorCaches bool // refresh CacheK / CacheV after every Forward

	logFn func(format string, v ...any)

	handle *C.lnb_ctx
	lt     *LlamaTransformer // keeps the transformer alive (and finalized after this context)
}

// LayerProgress switches the per-layer "Transformer block layer %d / %d was run" message (llamatransformer.go:163) on or off for
// contexts that were given a logFn.  The hook makes the library synchronise its stream after every block (that is what it times), which
// costs a few percent of a decode step: a host that only wants the text can set this to false.
var LayerProgress = true

// NewInferenceContext: same signature and defaults as src/model/inferencecontext.go:17-46.  The device side is created on the first
// Forward (the reference's constructor has no error return, and the context does not know its transformer until then).
func NewInferenceContext(model *Model, inferenceArgs common.InferenceArgs, logFn func(format string, v ...any)) *InferenceContext {
	context := &InferenceContext{logFn: logFn}
	if inferenceArgs.SequenceLength > 0 {
		context.SequenceLength = inferenceArgs.SequenceLength
	} else {
		context.SequenceLength = model.ModelArgs.MaxSequenceLength
	}
	modelArgs := model.ModelArgs
	context.CacheK = make([]*ml.Tensor, modelArgs.N_Layers)
	context.CacheV = make([]*ml.Tensor, modelArgs.N_Layers)
	for layerIdx := 0; layerIdx < modelArgs.N_Layers; layerIdx++ {
		context.CacheK[layerIdx], _ = ml.Zeros([]int{context.SequenceLength, modelArgs.N_KVHeads, modelArgs.HeadDim}, ml.DT_BF16)
		context.CacheV[layerIdx], _ = ml.Zeros([]int{context.SequenceLength, modelArgs.N_KVHeads, modelArgs.HeadDim}, ml.DT_BF16)
	}
	common.GLogger.DebugPrintf("Inference Context created with SequenceLength: %d", context.SequenceLength)
	return context
}

func (ic *InferenceContext) Logf(format string, v ...any) {
	if ic.logFn != nil {
		ic.logFn(format, v...)
	}
}

// The library's per-layer hook (lnb_ctx_set_layer_callback) lands here and becomes the reference's
// infContext.Logf("Transformer block layer %d / %d was run, took %.4f sec(s)", ...) of llamatransformer.go:163.
//
//export lnbGoLayerCallback
func lnbGoLayerCallback(layer C.int, nLayers C.int, secs C.double, user unsafe.Pointer) {
	ic := cgo.Handle(uintptr(user)).Value().(*InferenceContext)
	ic.Logf("Transformer block layer %d / %d was run, took %.4f sec(s)", int(layer), int(nLayers), float64(secs))
}

var queueWarned bool // (guarded by lt.mu's critical section order: a benign race at worst prints the line twice)

// RuntimeInfo reports what lnb_runtime_info says about the process: hardware queues the HIP runtime was told to use (and, with probe, how many
// streams really run concurrently), whether the library's default arrived before HIP initialised, device and clocks.
type RuntimeInfo struct {
	ABIVersion, Device, NumCUs                             int
	ShaderClockKHz, MemoryClockKHz, WallClockKHz           int
	HWQueuesEnv, HWQueuesExpected, HWQueuesMeasured        int
	HWQueuesSetByLibrary, HIPInitialisedBeforeLibraryLoad  bool
	DeviceName, Arch                                       string
}

func GetRuntimeInfo(device int, probeQueues bool) (RuntimeInfo, error) {
	var ri C.lnb_runtime_info_t
	probe := C.int(0)
	if probeQueues {
		probe = 1
	}
	if err := lnbCall(func() C.int { return C.lnb_runtime_info(C.int(device), probe, &ri) }); err != nil {
		return RuntimeInfo{}, err
	}
	return RuntimeInfo{ABIVersion: int(ri.abi_version), Device: int(ri.device), NumCUs: int(ri.n_cus),
		ShaderClockKHz: int(ri.shader_clock_khz), MemoryClockKHz: int(ri.memory_clock_khz), WallClockKHz: int(ri.wall_clock_khz),
		HWQueuesEnv: int(ri.hw_queues_env), HWQueuesExpected: int(ri.hw_queues_expected), HWQueuesMeasured: int(ri.hw_queues_measured),
		HWQueuesSetByLibrary: ri.hw_queues_set_by_library != 0, HIPInitialisedBeforeLibraryLoad: ri.hip_initialised_before_load != 0,
		DeviceName: C.GoString(&ri.device_name[0]), Arch: C.GoString(&ri.arch[0])}, nil
}

// attach creates the device-side context on the first Forward.
func (ic *InferenceContext) attach(lt *LlamaTransformer) error {
	if ic.handle != nil {
		return nil
	}
	if err := lnbCall(func() C.int { return C.lnb_ctx_create(lt.handle, C.int(ic.SequenceLength), &ic.handle) }); err != nil {
		return err
	}
	lt.mu.Lock()
	lt.ctxs++
	nctx := lt.ctxs
	lt.mu.Unlock()
	ic.lt = lt
	// One InferenceContext per generation, one goroutine each (inference.go:163-174): every context owns a HIP stream, and streams that
	// share a hardware queue run one after the other.  Say so ONCE when more contexts are alive than the runtime has queues.
	if nctx > 1 {
		var ri C.lnb_runtime_info_t
		if C.lnb_runtime_info(0, 0, &ri) == 0 && nctx > int(ri.hw_queues_expected) && !queueWarned {
			queueWarned = true
			why := ""
			if ri.hip_initialised_before_load != 0 && ri.hw_queues_set_by_library != 0 {
				why = " (HIP was initialised before liblnb_hip.so was loaded: export GPU_MAX_HW_QUEUES=16 before the process starts)"
			}
			common.GLogger.ConsolePrintf("warning: %d inference contexts on %d hardware queues: their steps will serialise%s", nctx, int(ri.hw_queues_expected), why)
		}
	}
	// Nothing pins the Go object: the reference creates one context per generation and never closes it (inference.go:174), so the
	// finalizer is what returns the device KV cache.  (A cgo.Handle held for the life of the context would keep it reachable for ever.)
	runtime.SetFinalizer(ic, func(c *InferenceContext) { c.Close() })
	return nil
}

// installLayerHook points the library's per-layer callback at this context for the duration of ONE Forward call: the cgo.Handle that
// lets the C side find the Go object exists only while the call runs (the callback fires synchronously inside lnb_forward, on the
// calling thread), so it never keeps the context alive.  The returned function removes the hook and deletes the handle.
func (ic *InferenceContext) installLayerHook() (release func(), err error) {
	if ic.logFn == nil || !LayerProgress {
		return func() {}, nil
	}
	h := cgo.NewHandle(ic)
	if err := lnbCall(func() C.int {
		return C.lnb_ctx_set_layer_callback(ic.handle, C.lnb_layer_cb(C.lnbGoLayerCallback), unsafe.Pointer(uintptr(h)))
	}); err != nil {
		h.Delete()
		return nil, err
	}
	return func() {
		C.lnb_ctx_set_layer_callback(ic.handle, nil, nil)
		h.Delete()
	}, nil
}

// SyncCachesFromDevice copies every layer's K and V cache into the host mirrors, in the reference's [position, kv head, dim] order
// (the library stores K position-contiguous on the device and hands it back transposed: lnb_ctx_read_kv).
func (ic *InferenceContext) SyncCachesFromDevice() error {
	if ic.handle == nil {
		return nil
	}
	for layer := range ic.CacheK {
		for which, t := range []*ml.Tensor{ic.CacheK[layer], ic.CacheV[layer]} {
			if err := lnbCall(func() C.int {
				return C.lnb_ctx_read_kv(ic.handle, C.int(layer), C.int(which), (*C.uint16_t)(unsafe.Pointer(&t.RawData[0])))
			}); err != nil {
				return err
			}
		}
	}
	return nil
}

// SetStopTokenIds hands model.Vocabulary.StopTokenIds to the device (lnb_ctx_set_stop_ids): the greedy loop then ends ON THE DEVICE with the
// first stop token (which is emitted, as inference.go:233-252 does), however many steps were enqueued behind it.  Up to 8 ids.
func (ic *InferenceContext) SetStopTokenIds(lt *LlamaTransformer, ids []TokenId) error {
	if err := ic.attach(lt); err != nil {
		return err
	}
	var p *C.int32_t
	if len(ids) > 0 {
		p = (*C.int32_t)(unsafe.Pointer(&ids[0]))
	}
	return lnbCall(func() C.int { return C.lnb_ctx_set_stop_ids(ic.handle, p, C.int(len(ids))) })
}

// DecodeGreedyUntil is the decode half of generateTokensInternal (inference.go:194-252) as ONE call: starting from `token` at position
// startPos it runs up to maxSteps Forward(1 token)+Argmax steps on the device and returns the tokens generated -- maxSteps of them unless a
// stop id ended the run (finished == true; the stop token is the last one).  The host loop that feeds generatedTokensCh calls it in chunks
// of any size: the tokens do not depend on the chunking.
func (ic *InferenceContext) DecodeGreedyUntil(lt *LlamaTransformer, token TokenId, startPos int, maxSteps int) (tokens []TokenId, finished bool, err error) {
	if maxSteps <= 0 { // &out[0] of an empty slice panics before the library can refuse the call
		return nil, false, fmt.Errorf("n_steps must be positive")
	}
	if err = ic.attach(lt); err != nil {
		return nil, false, err
	}
	out := make([]TokenId, maxSteps)
	var n, fin C.int
	if err = lnbCall(func() C.int {
		return C.lnb_decode_greedy_until(ic.handle, C.int32_t(token), C.int(startPos), C.int(maxSteps), (*C.int32_t)(unsafe.Pointer(&out[0])), &n, &fin, nil)
	}); err != nil {
		return nil, false, err
	}
	return out[:int(n)], fin != 0, nil
}

// SetThroughputSchedule selects the co-residency-friendly forms of the one-token kernels (lnb_ctx_set_schedule): for hosts that keep several
// generations in flight on one GPU, one InferenceContext each (inference.go:174).  Same tokens either way.
func (ic *InferenceContext) SetThroughputSchedule(lt *LlamaTransformer, on bool) error {
	if err := ic.attach(lt); err != nil {
		return err
	}
	s := C.int(C.LNB_SCHED_LATENCY)
	if on {
		s = C.int(C.LNB_SCHED_THROUGHPUT)
	}
	return lnbCall(func() C.int { return C.lnb_ctx_set_schedule(ic.handle, s) })
}

// Close frees the device buffers of this context (idempotent; also run by the finalizer, before the transformer's).  The library refuses to
// destroy a context that a live batch still holds (its tables and captured graphs keep the device pointers): the handle is kept then.
func (ic *InferenceContext) Close() error {
	if ic.handle == nil {
		return nil
	}
	if err := lnbCall(func() C.int { return C.lnb_ctx_destroy(ic.handle) }); err != nil {
		return err
	}
	ic.handle = nil
	ic.lt.mu.Lock()
	ic.lt.ctxs--
	ic.lt.mu.Unlock()
	return nil
}
