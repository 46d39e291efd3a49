#!/bin/bash
# Does MIOpen's find result survive the process?  One gpurun call: cold search, then two fresh processes on the same
# user-db / cache directories (round 4, DESIGN.md 7 "BM4DNet").  Run from the repo root.
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_probe/db MIOPEN_CUSTOM_CACHE_DIR=$PWD/gpurun_out/miopen_probe/cache
mkdir -p "$MIOPEN_USER_DB_PATH" "$MIOPEN_CUSTOM_CACHE_DIR"
echo "== cold: NDHWC + benchmark"; timeout -k 10 300 python tools/dbg/miopen_modes.py - 1 1
echo "== warm: NDHWC + benchmark again"; timeout -k 10 300 python tools/dbg/miopen_modes.py - 1 1
echo "== warm: NDHWC, benchmark off"; timeout -k 10 300 python tools/dbg/miopen_modes.py - 1 0
echo "== warm: default layout, benchmark off"; timeout -k 10 300 python tools/dbg/miopen_modes.py - 0 0
du -sh "$MIOPEN_USER_DB_PATH" "$MIOPEN_CUSTOM_CACHE_DIR"; ls "$MIOPEN_USER_DB_PATH" | head
