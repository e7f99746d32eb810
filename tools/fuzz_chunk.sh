#!/bin/bash
# Builds the page-header reader (host code of chunk.hip) with the address and undefined-behaviour
# sanitizers and fuzzes it with damaged copies of the column chunks in tests/golden/chunks.npz.
# CPU only:  bash tools/fuzz_chunk.sh [rounds per chunk] [seed]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/build_exp/fuzz_chunk
mkdir -p $B/chunks
python - "$R" "$B/chunks" <<'PY'
import sys, numpy as np
z = np.load(sys.argv[1] + "/tests/golden/chunks.npz")
for k, name in enumerate(n for n in z.files if n.endswith(":chunk")):
    z[name].tofile(f"{sys.argv[2]}/{k:03d}.bin")
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O1 -g -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
    -Xarch_host -fno-sanitize-recover=undefined $R/parquet-go_amd/csrc/chunk.hip $R/tools/fuzz_chunk.cpp -o $B/fuzz_chunk
$B/fuzz_chunk ${1:-200} ${2:-1} $B/chunks/*.bin
