#!/bin/bash
# copy the working tree to /tmp/t4snap and run pytest there, so that a long test run is not disturbed by edits (development aid)
set -e
D=${T4_SNAP_DIR:-/tmp/t4snap}   # (a second run beside a long one: T4_SNAP_DIR=/tmp/t4snap2)
rm -rf "$D"; mkdir -p "$D"
tar --exclude=.git --exclude=gpurun_out --exclude=.pytest_cache -cf - . | tar -xf - -C "$D"
cd "$D" && exec python -m pytest "$@"
