# The HIGSFA_DIAG build of the CURRENT tree (stamped instantiations, HIGSFA_WHATIF timing experiments) as tools/ab/libhigsfa_diag.so, out of tree:
#   bash tools/build_diag_lib.sh     then     HIGSFA_LIB=tools/ab/libhigsfa_diag.so python bench.py ...   (tools/whatif_nocopy.sh)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /tmp/hgdiag.XXXX)
cp -r "$R/pyfaceanalysis_amd" "$R/include" "$D/"
rm -rf "$D/pyfaceanalysis_amd/csrc/_obj" "$D/pyfaceanalysis_amd/libhigsfa.so"
(cd "$D" && HIGSFA_DIAG=1 python -m pyfaceanalysis_amd.build > "$D/build.log" 2>&1) || { tail -20 "$D/build.log"; exit 1; }
mkdir -p "$R/tools/ab"
cp "$D/pyfaceanalysis_amd/libhigsfa.so" "$R/tools/ab/libhigsfa_diag.so"
rm -rf "$D"
ls -la "$R/tools/ab/libhigsfa_diag.so"
