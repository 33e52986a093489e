// gpu.go — cgo shim between maxim2266/csvplus and libcsvplus_hip (include/csvplus_hip.h).
//
// Drop this file and join.go into the csvplus package directory (they are `package csvplus` and use its unexported
// names: Row, Index, indexImpl, mergeRows).  Build:
//
//	CGO_CFLAGS="-I$REPO/include" \
//	CGO_LDFLAGS="-L$REPO/csvplus_amd/lib -lcsvplus_hip -Wl,-rpath,$REPO/csvplus_amd/lib" go build
//
// NOT COMPILED IN THIS REPOSITORY: the build image has no Go toolchain (`go version`: command not found).  The same logic
// is compiled and tested in C++ (csvplus_amd/host/csvplus.hpp, tests/cpp/test_host.cpp: the reference's own tests through
// the C ABI); this file is the source a maintainer vets (gofmt / go vet) and adapts.  Line numbers refer to csvplus.go of
// the reference.
package csvplus

/*
#include <stdlib.h>
#include "csvplus_hip.h"
*/
import "C"

import (
	"errors"
	"runtime"
	"sync"
	"unsafe"
)

// One context per process.  A cph_ctx is single-threaded, so calls are serialised by gpuMu; the library calls
// hipSetDevice on every entry, so it does not matter which OS thread a goroutine runs on.
var (
	gpuOnce sync.Once
	gpuCtx  *C.cph_ctx
	gpuMu   sync.Mutex
	gpuErr  error
)

func gpu() (*C.cph_ctx, error) {
	gpuOnce.Do(func() {
		if rc := C.cph_ctx_create(0, &gpuCtx); rc != C.CPH_OK {
			gpuErr = errors.New("csvplus: no usable GPU (libcsvplus_hip has no CPU fallback)")
		}
	})
	return gpuCtx, gpuErr
}

func lastErr(ctx *C.cph_ctx) error { return errors.New(C.GoString(C.cph_last_error(ctx))) }

// pinnedBuf is one block of C-owned pinned host memory that grows and is reused: cgo must not hand Go-heap pointers that C
// retains, pinned memory makes the H2D copy asynchronous, and page-locking costs far more than a probe of 8192 rows — so a
// batch must not allocate.  Guarded by gpuMu.
type pinnedBuf struct {
	p   unsafe.Pointer
	cap int
}

func (b *pinnedBuf) need(ctx *C.cph_ctx, n int) (unsafe.Pointer, error) {
	if n <= b.cap {
		return b.p, nil
	}
	if b.p != nil {
		C.cph_pinned_free(ctx, b.p)
		b.p, b.cap = nil, 0
	}
	want := n + n/2 + 64
	if C.cph_pinned_alloc(ctx, C.size_t(want), &b.p) != C.CPH_OK {
		return nil, lastErr(ctx)
	}
	b.cap = want
	return b.p, nil
}

// staging holds the SoA form of up to maxStageCols key columns of a batch of rows: per column one data block and one
// offsets block, kept between batches.
const maxStageCols = 4 * C.CPH_MAX_KEY_COLS

type staging struct {
	data, offs [maxStageCols]pinnedBuf
	used       int
}

var stagePool staging // guarded by gpuMu

// reset starts a new batch (the blocks of the previous one are reused).
func (s *staging) reset() { s.used = 0 }

// stage appends the columns `columns` of `rows` and returns their descriptors (valid until the next reset).
func (s *staging) stage(ctx *C.cph_ctx, rows []Row, columns []string) ([]C.cph_strcol, error) {
	cols := make([]C.cph_strcol, len(columns))
	for c, name := range columns {
		if s.used == maxStageCols {
			return nil, errors.New("csvplus: too many key columns in one device call")
		}
		total := 0
		for _, r := range rows {
			total += len(r[name])
		}
		data, err := s.data[s.used].need(ctx, total+8)
		if err != nil {
			return nil, err
		}
		offs, err := s.offs[s.used].need(ctx, 8*(len(rows)+1))
		if err != nil {
			return nil, err
		}
		s.used++
		d := unsafe.Slice((*byte)(data), total+8)
		o := unsafe.Slice((*uint64)(offs), len(rows)+1)
		pos := 0
		for i, r := range rows {
			o[i] = uint64(pos)
			pos += copy(d[pos:], r[name])
		}
		o[len(rows)] = uint64(pos)
		cols[c] = C.cph_strcol{data: (*C.uint8_t)(data), offsets: offs, nrows: C.uint64_t(len(rows)),
			offset_bits: 64, mem: C.CPH_MEM_HOST}
	}
	return cols, nil
}

// gpuIndex keeps the device twin of index.impl.rows' key columns alive (a field `gpu *gpuIndex` of Index).
// side: columns of impl.rows staged once, in SORTED order (row i = impl.rows[i]), for chains whose later key is a column of
// THIS index's rows (cph_chain_step.source < 0: join.go columnOrigin / fusedBatch); a nil entry = some row lacks the column.
// presence: per column name, how many rows of impl.rows carry it (all / none / some) — one walk over the rows per column,
// then the answer is a map lookup per batch (join.go columnOrigin asks once per 8192 stream rows and earlier index).
// Both maps die with the gpuIndex: whatever replaces impl.rows (ResolveDuplicates) builds a new device twin.
type gpuIndex struct {
	h        *C.cph_index
	side     map[string]*sideColumn
	presence map[string]int8
}

const (
	colNone int8 = iota
	colSome
	colAll
)

// columnPresence: colAll / colNone / colSome for column `name` over index.impl.rows, cached.  Call with gpuMu held.
func (index *Index) columnPresence(name string) int8 {
	g := index.gpu
	if g.presence == nil {
		g.presence = map[string]int8{}
	}
	if p, seen := g.presence[name]; seen {
		return p
	}
	have := 0
	for _, r := range index.impl.rows {
		if _, ok := r[name]; ok {
			have++
		}
	}
	p := colSome
	if have == 0 {
		p = colNone
	} else if have == len(index.impl.rows) {
		p = colAll
	}
	g.presence[name] = p
	return p
}

// sideColumn is one column of an index's rows as pinned SoA (its blocks live as long as the index: they are freed by the
// index's finalizer, not reused between batches like the stagePool).
type sideColumn struct {
	data, offs pinnedBuf
	col        C.cph_strcol
}

// sideColumn stages column `name` of index.impl.rows (sorted order) on first use.  Call with gpuMu held.
func (index *Index) sideColumn(ctx *C.cph_ctx, name string) (*C.cph_strcol, error) {
	g := index.gpu
	if g.side == nil {
		g.side = map[string]*sideColumn{}
	}
	if sc, seen := g.side[name]; seen {
		if sc == nil {
			return nil, nil
		}
		return &sc.col, nil
	}
	rows := index.impl.rows
	if index.columnPresence(name) != colAll {
		g.side[name] = nil // (mergeRows would then take the column from further down the chain, or miss it)
		return nil, nil
	}
	total := 0
	for _, r := range rows {
		total += len(r[name])
	}
	sc := &sideColumn{}
	data, err := sc.data.need(ctx, total+8)
	if err != nil {
		return nil, err
	}
	offs, err := sc.offs.need(ctx, 8*(len(rows)+1))
	if err != nil {
		return nil, err
	}
	d := unsafe.Slice((*byte)(data), total+8)
	o := unsafe.Slice((*uint64)(offs), len(rows)+1)
	pos := 0
	for i, r := range rows {
		o[i] = uint64(pos)
		pos += copy(d[pos:], r[name])
	}
	o[len(rows)] = uint64(pos)
	sc.col = C.cph_strcol{data: (*C.uint8_t)(data), offsets: offs, nrows: C.uint64_t(len(rows)), offset_bits: 64, mem: C.CPH_MEM_HOST}
	g.side[name] = sc
	return &sc.col, nil
}

// sortOnGPU replaces `sort.Sort(&index.impl)` (csvplus.go:736) and, for unique indices, the adjacent-equal scan of
// createUniqueIndex (:749-753).  impl.rows come back sorted (stable: rows with equal keys keep their input order, one of
// the orders the reference's unstable sort may produce).  Returns the sorted position of the first row equal to its
// predecessor, or -1.
func sortOnGPU(impl *indexImpl, unique bool) (*gpuIndex, int, error) {
	ctx, err := gpu()
	if err != nil {
		return nil, -1, err
	}
	if uint64(len(impl.rows)) > 0xFFFFFFFF {
		return nil, -1, errors.New("csvplus: more than 2^32-1 rows in a GPU index")
	}
	gpuMu.Lock()
	defer gpuMu.Unlock()
	stagePool.reset()
	cols, err := stagePool.stage(ctx, impl.rows, impl.columns)
	if err != nil {
		return nil, -1, err
	}
	var h *C.cph_index
	var dup C.uint64_t
	u := C.int32_t(0)
	if unique {
		u = 1
	}
	rc := C.cph_index_build(ctx, &cols[0], C.int32_t(len(cols)), u, &h, &dup)
	if rc != C.CPH_OK && rc != C.CPH_ERR_DUPLICATE {
		return nil, -1, lastErr(ctx)
	}
	var perm *C.uint32_t
	var n C.uint64_t
	if C.cph_index_perm(h, C.CPH_MEM_HOST, &perm, &n) != C.CPH_OK {
		C.cph_index_destroy(h)
		return nil, -1, lastErr(ctx)
	}
	p := unsafe.Slice((*uint32)(unsafe.Pointer(perm)), int(n))
	sorted := make([]Row, len(impl.rows))
	for i, src := range p {
		sorted[i] = impl.rows[src]
	}
	impl.rows = sorted // from here on a sorted POSITION is the subscript into impl.rows, as in the reference
	if rc == C.CPH_ERR_DUPLICATE {
		C.cph_index_destroy(h) // the reference returns a nil index (:751); the caller formats the error from rows[dup]
		return nil, int(dup), nil
	}
	g := &gpuIndex{h: h}
	runtime.SetFinalizer(g, func(g *gpuIndex) {
		gpuMu.Lock()
		C.cph_index_destroy(g.h)
		for _, sc := range g.side {
			if sc != nil {
				C.cph_pinned_free(gpuCtx, sc.data.p)
				C.cph_pinned_free(gpuCtx, sc.offs.p)
			}
		}
		gpuMu.Unlock()
	})
	return g, -1, nil
}

// createIndex (:707-738) then ends with
//
//	gi, dup, err := sortOnGPU(&index.impl, unique)
//	if err != nil { return nil, err }
//	if dup >= 0 {   // :751, the reference's text
//		return nil, errors.New("duplicate value while creating unique index: " +
//			index.impl.rows[dup].SelectExisting(columns...).String())
//	}
//	index.gpu = gi

// find replaces indexImpl.find (:870-891): [lower, upper) over impl.rows.
func (index *Index) findOnGPU(values []string) (int, int, error) {
	ctx, err := gpu()
	if err != nil {
		return 0, 0, err
	}
	if len(values) == 0 {
		return 0, len(index.impl.rows), nil // :872-874
	}
	vals := make([]C.cph_strval, len(values))
	pin := make([]unsafe.Pointer, len(values))
	for i, v := range values {
		pin[i] = C.CBytes([]byte(v))
		vals[i] = C.cph_strval{data: (*C.uint8_t)(pin[i]), len: C.uint64_t(len(v))}
	}
	defer func() {
		for _, p := range pin {
			C.free(p)
		}
	}()
	var lo, hi C.uint64_t
	gpuMu.Lock()
	defer gpuMu.Unlock()
	if C.cph_index_find(ctx, index.gpu.h, &vals[0], C.int32_t(len(vals)), &lo, &hi) != C.CPH_OK {
		return 0, 0, lastErr(ctx)
	}
	return int(lo), int(hi), nil
}
