//go:build hip

package model

import "testing"

// TestDeviceExpAgainstMathExp: the device's float64 exp against math.Exp on every possible softmax input (see exptable_hip.go).  Both are faithful
// implementations: more than one ulp apart anywhere would be a defect on one side; the count of one-ulp differences is what the reference's own
// arithmetic is away from the device's before the float32 narrowing of operations_impl.go:506 absorbs it.
func TestDeviceExpAgainstMathExp(t *testing.T) {
	for _, divisor := range []float32{1.0, 11.3125} {
		differ, maxUlps, err := ExpDistanceFromGo(0, divisor)
		if err != nil {
			t.Fatal(err)
		}
		t.Logf("divisor %v: %d of 65536 inputs differ from math.Exp, at most %d float64 ulp(s)", divisor, differ, maxUlps)
		if maxUlps > 1 {
			t.Errorf("divisor %v: %d ulps apart", divisor, maxUlps)
		}
	}
}
