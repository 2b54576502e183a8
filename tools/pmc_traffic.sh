# HBM bytes per launch position for one or more prebuilt libraries on ONE box (separate --pmc passes, MI355X_MICROARCH.md §HBM):
#   bash tools/pmc_traffic.sh <tag> lib1 [lib2 ...]      ("-" = the in-tree build)  ->  gpurun_out/<tag>_traffic.txt
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/${TAG}_traffic.txt
i=0
for L in "$@"; do
  i=$((i+1))
  if [ "$L" = "-" ]; then unset HIGSFA_LIB; else export HIGSFA_LIB=$R/$L; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_${i}_$C -- python3 $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-frame --no-inflight --no-extra-legs > $R/gpurun_out/pmc_${TAG}_${i}_$C.log 2>&1 || { echo "pass $C of $L failed"; tail -3 $R/gpurun_out/pmc_${TAG}_${i}_$C.log; exit 1; }
  done
  python3 $R/tools/pmc_traffic.py "$L" $R/gpurun_out/pmc_${TAG}_${i}_FETCH_SIZE $R/gpurun_out/pmc_${TAG}_${i}_WRITE_SIZE >> $R/gpurun_out/${TAG}_traffic.txt
  rm -rf $R/gpurun_out/pmc_${TAG}_${i}_FETCH_SIZE $R/gpurun_out/pmc_${TAG}_${i}_WRITE_SIZE
done
cat $R/gpurun_out/${TAG}_traffic.txt
