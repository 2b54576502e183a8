# A/B of two builds on ONE box without rebuilding there: build the library of a given commit (default HEAD) out of tree and keep it as
# tools/ab/libhigsfa_<tag>.so (git-ignored, but it travels to the GPU box); `HIGSFA_LIB=tools/ab/libhigsfa_<tag>.so python bench.py ...`
# then runs that build (pyfaceanalysis_amd/_capi.py).   bash tools/build_ref_lib.sh [commit] [tag]
set -e
C=${1:-HEAD}; TAG=${2:-base}
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /tmp/hgref.XXXX)
git -C "$R" archive "$C" pyfaceanalysis_amd include | tar -x -C "$D"
(cd "$D" && python -m pyfaceanalysis_amd.build > "$D/build.log" 2>&1) || { tail -20 "$D/build.log"; exit 1; }
mkdir -p "$R/tools/ab"
cp "$D/pyfaceanalysis_amd/libhigsfa.so" "$R/tools/ab/libhigsfa_$TAG.so"
rm -rf "$D"
ls -la "$R/tools/ab/libhigsfa_$TAG.so"
