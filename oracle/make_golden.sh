#!/bin/bash
# Regenerate tests/golden/*.npz from the real reference (optional arguments: names of the fixture files to write).  Build container only
# (needs /root/reference and /opt/conda/bin/python3.9 = numpy 1.26 / scipy 1.7 /
# h5py 3.3, the closest available match to the reference's lock file).
# TEST INFRASTRUCTURE ONLY.
set -euo pipefail
here="$(cd "$(dirname "$0")/.." && pwd)"
tmp="$(mktemp -d)"
trap 'rm -rf "$tmp"' EXIT
# the reference imports these two optional packages at module load and never
# uses them on this path (computational_routine.py:25, specest/fooofspy.py:10)
mkdir -p "$tmp/stubs/dask_jobqueue" "$tmp/stubs/fooof" "$tmp/spydir"
echo "class SLURMCluster: pass" > "$tmp/stubs/dask_jobqueue/__init__.py"
echo "class FOOOF: pass" > "$tmp/stubs/fooof/__init__.py"
cd "$tmp"
if [ "${1:-}" = "--container" ]; then      # the .spy container fixtures (tests/golden/spy_container)
  SPYDIR="$tmp/spydir" SPYSILENTSTARTUP=1 SPYLOGLEVEL=ERROR PYTHONPATH="$tmp/stubs:/root/reference" \
    /opt/conda/bin/python3.9 -W ignore "$here/oracle/gen_spy_container.py" "$here/tests/golden/spy_container"
  exit 0
fi
if [ "${1:-}" = "--check-container" ]; then   # load a container written by syncopy_amd.io.save with the reference
  SPYDIR="$tmp/spydir" SPYSILENTSTARTUP=1 SPYLOGLEVEL=ERROR PYTHONPATH="$tmp/stubs:/root/reference" \
    /opt/conda/bin/python3.9 -W ignore "$here/oracle/gen_spy_container.py" --check "$2"
  exit 0
fi
SPYDIR="$tmp/spydir" SPYSILENTSTARTUP=1 SPYLOGLEVEL=ERROR \
PYTHONPATH="$tmp/stubs:/root/reference" \
  /opt/conda/bin/python3.9 -W ignore "$here/oracle/gen_golden.py" "$here/tests/golden" "$@"
