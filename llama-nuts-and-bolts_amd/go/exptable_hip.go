//go:build hip

// exptable_hip.go -- how far is the DEVICE's softmax exp from Go's math.Exp?  (package model, `-tags hip`; never compiled in this repository: no Go toolchain in the image.)
//
// The reference computes the softmax numerator as math.Exp(float64(x)) on a BFloat16 score (src/ml/operations_impl.go:498, after DivToScalar at
// src/model/llamatransformer.go:464): a function of 16 bits.  lnb_op_exp_table returns the device's value for every one of the 65536 possible raw scores, so the distance
// between the two implementations can be measured exhaustively instead of argued.  This repository measured the same against glibc's exp (what its CPU oracle calls):
// 231 inputs differ, each by one float64 ulp, none after narrowing to float32 (tests/test_gpu_parity.py).  math.Exp is a third implementation; run
// TestDeviceExpAgainstMathExp (exptable_hip_test.go) on a machine with Go and an MI355X to learn its number.

package model

/*
#cgo LDFLAGS: -llnb_hip
#include <stdint.h>
#include "lnb.h"
*/
import "C"

import (
	"errors"
	"math"
	"unsafe"
)

// DeviceExpTable returns out[s] = exp(float64(truncBF16(float32(bf16 s) / divisor))) for s = 0 .. 65535 as the attention kernels evaluate it.
func DeviceExpTable(device int, divisor float32) ([]float64, error) {
	out := make([]float64, 1<<16)
	if C.lnb_op_exp_table(C.int(device), C.float(divisor), (*C.double)(unsafe.Pointer(&out[0]))) != 0 {
		return nil, errors.New(C.GoString(C.lnb_last_error()))
	}
	return out, nil
}

// ExpDistanceFromGo compares DeviceExpTable with math.Exp over all 65536 inputs: how many differ and by how many float64 ulps at most
// (NaN inputs must be NaN on both sides; +Inf / 0 must coincide).
func ExpDistanceFromGo(device int, divisor float32) (differ int, maxUlps uint64, err error) {
	dev, err := DeviceExpTable(device, divisor)
	if err != nil {
		return 0, 0, err
	}
	for s := 0; s < 1<<16; s++ {
		x := math.Float32frombits(uint32(s) << 16)
		q := x / divisor                                             // float32 division (ml.DivToScalar on a float32 item)
		t := math.Float32frombits(math.Float32bits(q) & 0xFFFF0000) // dtype.BFloat16fromFloat32: truncation (src/dtype/bfloat16.go:31-33)
		want := math.Exp(float64(t))
		got := dev[s]
		if math.IsNaN(want) || math.IsNaN(got) {
			if math.IsNaN(want) != math.IsNaN(got) {
				return differ, maxUlps, errors.New("NaN on one side only")
			}
			continue
		}
		a, b := math.Float64bits(got), math.Float64bits(want)
		if a != b {
			differ++
			d := a - b
			if b > a {
				d = b - a
			}
			if d > maxUlps {
				maxUlps = d
			}
		}
	}
	return differ, maxUlps, nil
}
