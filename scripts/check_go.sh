#!/bin/bash
# Type-checks the Go binding (llama-nuts-and-bolts_amd/go/*_hip.go) against a checkout of adalkiran/llama-nuts-and-bolts when a Go
# toolchain exists; a no-op (exit 0) otherwise.  The build image of this repository has no Go toolchain, so CI cannot run it.
#   scripts/check_go.sh [/path/to/llama-nuts-and-bolts checkout]      (default: $LNB_REFERENCE or /root/reference)
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
REF="${1:-${LNB_REFERENCE:-/root/reference}}"
if ! command -v go >/dev/null 2>&1; then echo "check_go: no Go toolchain on this machine -- skipped"; exit 0; fi
if [ ! -f "$REF/go.mod" ]; then echo "check_go: no checkout of the reference at $REF -- skipped"; exit 0; fi
W="$(mktemp -d)"; trap 'rm -rf "$W"' EXIT
cp -r "$REF/." "$W/"
# the two reference files the binding replaces (and the test that pokes their private fields) step aside under -tags hip
for f in src/model/llamatransformer.go src/model/inferencecontext.go src/model/llamatransformer_simulated_test.go; do
    [ -f "$W/$f" ] && sed -i '1i //go:build !hip\n' "$W/$f"
done
cp "$HERE"/llama-nuts-and-bolts_amd/go/*_hip.go "$W/src/model/"
cd "$W"
export CGO_ENABLED=1 CGO_CFLAGS="-I$HERE/include" CGO_LDFLAGS="-L$HERE/llama-nuts-and-bolts_amd -llnb_hip -Wl,-rpath,$HERE/llama-nuts-and-bolts_amd"
go vet -tags hip ./src/model/ ./src/inference/ && go build -tags hip ./... && echo "check_go: ok (model.go:48 and inference.go:202 compile unchanged against the hip files)"
