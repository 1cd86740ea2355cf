#!/bin/bash
# round 3, call R: weight loads of gemm_stream_kernel non-temporal vs default policy (time + FETCH_SIZE)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
L=gpurun_out/r03r_gemmstream.log; : > $L
for S in 128 512 4096; do for shape in "6144 4096 0 1" "4096 4096 0 1" "14336 4096 0 2" "4096 14336 0 1"; do for b in 0 nt0; do echo -n "weights_nt=$([ $b = 0 ] && echo 1 || echo 0) " >> $L; timeout 60 tools/gemmstream_bench_$b $S $shape >> $L 2>&1; done; done; done
sed 's/ lds=[0-9]*//; s/ err=no error//; s/occ=2\/2 R=3\/3 //; s/GS_DBG=0 //; s/ per launch//' $L
for b in 0 nt0; do for S in 128 4096; do
  P=$PWD/gpurun_out/prof_r03r_${b}_$S; mkdir -p $P
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P -o pmc -- $GRAFT_REPO_ROOT/tools/gemmstream_bench_$b $S 14336 4096 0 2 > /dev/null 2>&1 )
  python - <<PY
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("$P/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "gemm_stream" in r["Kernel_Name"]]
print("bench_$b gate|up S=$S: %d launches, fabric read per launch %.1f MB" % (len(v), 2 * sum(v) / max(1, len(v)) / 1024.0))
PY
done; done
