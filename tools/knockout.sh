#!/bin/bash
# Knock-out builds of the library (diagnostics): tools/ko/<name>/libfcaf3d_hip.so = the product library with conv.hip (exec.hip for
# FC_KO_EXEC) compiled under -D<flag>.  Python tools take one through FC_LIB=tools/ko/<name>/libfcaf3d_hip.so.  Run a tool against one with LD_LIBRARY_PATH=tools/ko/<name> (tools/nbench has a RUNPATH, the variable wins).
#   tools/knockout.sh FC_KO_WG_NOMFMA FC_KO_WG_NOSPLIT ...
set -e
cd "$(dirname "$0")/.."
python -c "from fcaf3d_amd.build import build; build(verbose=False)"
for f in "$@"; do
  ( d=tools/ko/$f; mkdir -p $d
    src=conv; case $f in FC_KO_EXEC*) src=exec;; esac
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result -D$f -c fcaf3d_amd/csrc/$src.hip -o $d/$src.o
    objs=$(ls fcaf3d_amd/csrc/*.o | grep -v "/$src.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libfcaf3d_hip.so $d/$src.o $objs ) &
done
wait
ls -la tools/ko/*/libfcaf3d_hip.so
