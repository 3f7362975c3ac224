#!/bin/bash
# syncopy_amd's compute functions under the reference's own engine (oracle/check_dropin.py).  Build container only: needs
# /root/reference, /opt/conda/bin/python3.9 (h5py / dask for the reference) and /usr/bin/python3 (torch for the product).
# Optional argument: a file to write the log to (profiles/r5_dropin_check.txt).  TEST INFRASTRUCTURE ONLY.
set -euo pipefail
here="$(cd "$(dirname "$0")/.." && pwd)"
tmp="$(mktemp -d)"
trap 'rm -rf "$tmp"' EXIT
mkdir -p "$tmp/stubs/dask_jobqueue" "$tmp/stubs/fooof" "$tmp/spydir"
echo "class SLURMCluster: pass" > "$tmp/stubs/dask_jobqueue/__init__.py"
echo "class FOOOF: pass" > "$tmp/stubs/fooof/__init__.py"
log="${1:-}"
[ -n "$log" ] && log="$(cd "$(dirname "$log")" && pwd)/$(basename "$log")"
cd "$tmp"
SPYDIR="$tmp/spydir" SPYSILENTSTARTUP=1 SPYLOGLEVEL=ERROR PYTHONPATH="$tmp/stubs:/root/reference" \
  /opt/conda/bin/python3.9 -W ignore "$here/oracle/check_dropin.py" $log
