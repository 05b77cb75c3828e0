#!/bin/sh
# make_overlay.sh <reference checkout> [<output json>]
#
# 1. copies the cgo binding into the checkout as the NEW directory <ref>/gdg (package github.com/andrepxx/go-dsp-guitar/gdg inside the
#    reference's own module).  It has to be a real directory: cmd/go runs cgo with the package directory as its working directory, so a
#    package that exists only as an overlay entry does not build.  No file of the reference is modified, go.mod is not touched.
# 2. writes the `go build -overlay` file for the three REPLACED files (signal, tuner, spatializer; pure Go, no cgo): keys are absolute
#    paths of files that exist in the checkout, values absolute paths inside this repository.
set -e
REF=$(cd "${1:?usage: make_overlay.sh <reference checkout> [<output json>]}" && pwd)
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${2:-/tmp/gdg-overlay.json}
mkdir -p "$REF/gdg"
cp "$HERE/gdg/gdg.go" "$REF/gdg/gdg.go"
cat > "$OUT" <<JSON
{
  "Replace": {
    "$REF/signal/signal.go": "$HERE/signal/signal.go",
    "$REF/tuner/tuner.go": "$HERE/tuner/tuner.go",
    "$REF/spatializer/spatializer.go": "$HERE/spatializer/spatializer.go"
  }
}
JSON
echo "$OUT"
