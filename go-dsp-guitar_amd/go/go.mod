// The module path of the reference (go.mod:1) and its language level (go.mod:3): `go vet ./gdg && go build ./gdg` compiles the cgo binding
// stand-alone against include/gdg.h (tests/test_go_build.py runs exactly that wherever a Go toolchain is on PATH).  The three overlay
// packages (signal, tuner, spatializer) import the reference's own packages and only build inside a reference checkout (INTEGRATION.md, section 3).
module github.com/andrepxx/go-dsp-guitar

go 1.16
