#!/bin/bash
# tools/build_variant.sh <name> <extra -D flags...> : experimental libtiddit_hip variants under gpurun-visible build/
set -e
NAME=$1; shift
SRC=${VARIANT_SRC:-tdt_coverage}
cd "$(dirname "$0")/.."
mkdir -p variants
OBJS=""
for f in tdt_ctx tdt_coverage tdt_gc tdt_dbscan tdt_dbscan_yseg tdt_sort tdt_bam tdt_signal tdt_median tdt_region tdt_bgzf tdt_inflate tdt_inflate2 tdt_ingest tdt_format tdt_comm tdt_means tdt_stats tdt_sigtab; do [ "$f" != "$SRC" ] && OBJS="$OBJS tiddit_amd/csrc/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c tiddit_amd/csrc/$SRC.hip -o variants/cov_$NAME.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS variants/cov_$NAME.o -o variants/lib_$NAME.so -lz -lpthread -ldl
echo built variants/lib_$NAME.so
