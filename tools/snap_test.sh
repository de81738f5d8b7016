#!/bin/bash
# copy the working tree to /tmp/t4snap and run pytest there, so that a long test run is not disturbed by edits (development aid)
set -e
rm -rf /tmp/t4snap; mkdir -p /tmp/t4snap
tar --exclude=.git --exclude=gpurun_out --exclude=.pytest_cache -cf - . | tar -xf - -C /tmp/t4snap
cd /tmp/t4snap && exec python -m pytest "$@"
