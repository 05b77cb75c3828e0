#!/bin/sh
# make_overlay.sh <reference checkout> [<output json>]
# Writes the `go build -overlay` file for the reference tree: the three overlaid packages (signal, tuner, spatializer) and the cgo
# binding, ADDED to the reference module as package github.com/andrepxx/go-dsp-guitar/gdg (an overlay may name files that do not exist
# on disk).  Keys are absolute paths inside the checkout, values absolute paths inside this repository: nothing is copied, go.mod is
# not touched.
set -e
REF=$(cd "${1:?usage: make_overlay.sh <reference checkout> [<output json>]}" && pwd)
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${2:-/tmp/gdg-overlay.json}
cat > "$OUT" <<JSON
{
  "Replace": {
    "$REF/signal/signal.go": "$HERE/signal/signal.go",
    "$REF/tuner/tuner.go": "$HERE/tuner/tuner.go",
    "$REF/spatializer/spatializer.go": "$HERE/spatializer/spatializer.go",
    "$REF/gdg/gdg.go": "$HERE/gdg/gdg.go"
  }
}
JSON
echo "$OUT"
