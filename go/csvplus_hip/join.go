// join.go — DataSource.Join / Except over libcsvplus_hip: the per-row first() + forward scan (csvplus.go:557-563) and
// has() (:599-602) become one device call per batch of stream rows; a chain of Joins over one stream becomes ONE fused call
// per batch (cph_join_chain_ex reporting sorted positions — the entry point bench.py times).
//
// NOT COMPILED IN THIS REPOSITORY (no Go toolchain in the build image); compiled and tested twin:
// csvplus_amd/host/csvplus.hpp (probeSource / chainSource / columnOrigin), tests/cpp/test_host.cpp (TestBatchingSemantics,
// TestChainPrecedence).  See gpu.go for the build line.
package csvplus

/*
#include "csvplus_hip.h"
*/
import "C"

import (
	"errors"
	"io"
	"unsafe"
)

// joinBatch rows are read ahead per device call.  1 reproduces the reference's row-at-a-time order of side effects
// exactly; larger batches only move upstream side effects earlier (SURVEY.md §8b): rows are still emitted in the reference's
// order, an error or io.EOF from fn stops the emission at once and discards the read-ahead, and a row that fails
// SelectValues surfaces its error only after the rows in front of it were delivered.
var joinBatch = 8192

// chainStep is one Join of a chain over a common stream.
type chainStep struct {
	index   *Index
	columns []string
}

type chainSpec struct {
	upstream DataSource
	steps    []chainStep
}

// Join (:545-569): one step.
func (src DataSource) Join(index *Index, columns ...string) DataSource {
	return src.JoinChain(JoinStep{index, columns})
}

// JoinStep is one Join of JoinChain: the index and the stream columns matched with its key columns (none: natural join).
type JoinStep struct {
	Index   *Index
	Columns []string
}

// JoinChain(s0, s1, ...) returns the rows of src.Join(s0.Index, s0.Columns...).Join(s1.Index, s1.Columns...)... — same
// rows, same order, same errors — with ONE device call per batch of stream rows for all steps (at most CPH_MAX_CHAIN = 8
// per call; longer chains are cut into several).  It is an ADDITION to the reference's method set: the reference's
// DataSource is a bare func type, so a Join cannot see that its source is another Join's closure (Go func values have no
// identity to look up) and `a.Join(x).Join(y)` written with the unchanged API stays two probes per batch.  A maintainer
// who turns DataSource into `struct { fn func(RowFunc) error; chain *chainSpec }` makes the fusion automatic, as the C++
// twin of this file does (csvplus_amd/host/csvplus.hpp: DataSource::Join).
func (src DataSource) JoinChain(steps ...JoinStep) DataSource {
	if len(steps) == 0 {
		return src
	}
	cut := len(steps)
	if cut > C.CPH_MAX_CHAIN {
		cut = C.CPH_MAX_CHAIN
	}
	spec := &chainSpec{upstream: src}
	for _, st := range steps[:cut] {
		columns := st.Columns
		if len(columns) == 0 {
			columns = st.Index.impl.columns // :546-547
		} else if len(columns) > len(st.Index.impl.columns) {
			panic("too many source columns in Join()") // :548-550
		}
		spec.steps = append(spec.steps, chainStep{st.Index, columns})
	}
	return chainSource(spec).JoinChain(steps[cut:]...)
}

// batched reads src in batches of rows that carry `columns` (SelectValues, :556) and hands each batch to flush.
func batched(src DataSource, columns []string, flush func(batch []Row) error) error {
	batch := make([]Row, 0, joinBatch)
	err := src(func(row Row) error {
		if _, e := row.SelectValues(columns...); e != nil { // rows before this one first, then the error
			fe := flush(batch)
			batch = batch[:0] // whatever flush said: these rows were handed over, they must not be emitted again
			if fe != nil {
				return fe
			}
			return e
		}
		batch = append(batch, row)
		if len(batch) >= joinBatch {
			fe := flush(batch)
			batch = batch[:0]
			return fe
		}
		return nil
	})
	if err != nil {
		return err
	}
	if err = flush(batch); err == io.EOF { // the source would have mapped it (:238-239)
		err = nil
	}
	return err
}

// probeBatch is one Join (anti == false) or Except (anti == true) over a batch whose rows all carry `columns`.
func probeBatch(index *Index, columns []string, anti bool, batch []Row, emit RowFunc) error {
	if len(batch) == 0 {
		return nil
	}
	ctx, err := gpu()
	if err != nil {
		return err
	}
	gpuMu.Lock()
	stagePool.reset()
	cols, err := stagePool.stage(ctx, batch, columns)
	if err != nil {
		gpuMu.Unlock()
		return err
	}
	var m *C.cph_matches
	rc := C.cph_join_probe(ctx, index.gpu.h, &cols[0], C.int32_t(len(columns)), nil, 32, 0, 0, 0,
		0 /* bounds only */, C.CPH_MEM_HOST, &m)
	if rc != C.CPH_OK {
		e := lastErr(ctx)
		gpuMu.Unlock()
		return e
	}
	lo := unsafe.Slice((*uint32)(unsafe.Pointer(m.lo)), len(batch))
	cnt := unsafe.Slice((*uint32)(unsafe.Pointer(m.cnt)), len(batch))
	gpuMu.Unlock()
	defer func() { gpuMu.Lock(); C.cph_matches_release(m); gpuMu.Unlock() }()
	for i, row := range batch { // stream order, then ascending index position (:559-563)
		if anti {
			if cnt[i] == 0 { // :600-602
				if err := emit(row); err != nil {
					return err
				}
			}
			continue
		}
		for j := uint32(0); j < cnt[i]; j++ {
			if err := emit(mergeRows(index.impl.rows[lo[i]+j], row)); err != nil { // impl.rows read live (:557)
				return err
			}
		}
	}
	return nil
}

// runSteps applies steps[from:] one after the other: step k's output rows are step k+1's stream rows, in emission order —
// the same order of outputs and of errors as the reference's nested closures.
func runSteps(spec *chainSpec, from int, rows []Row, fn RowFunc) error {
	cur := rows
	for k := from; k < len(spec.steps); k++ {
		st := spec.steps[k]
		if k > from { // SelectValues of this step over ITS stream rows (:556): the rows in front of a failing one still go through
			for i, r := range cur {
				if _, miss := r.SelectValues(st.columns...); miss != nil {
					if e := runSteps(spec, k, cur[:i], fn); e != nil {
						return e
					}
					return miss
				}
			}
		}
		emit := fn
		var next []Row
		if k+1 < len(spec.steps) {
			emit = func(row Row) error { next = append(next, row); return nil }
		}
		if err := probeBatch(st.index, st.columns, false, cur, emit); err != nil {
			return err
		}
		cur = next
	}
	return nil
}

// chainSource is upstream.Join(steps[0])...Join(steps[k]).  Per batch of stream rows: ONE fused device call when every
// row carries the key columns of ALL steps itself — mergeRows (:571-583) lets the right operand win, so the value a later
// Join sees is then the stream's own —, else the steps one after the other.  A step-k output row is
// mergeRows(index_k row, mergeRows(index_k-1 row, ... stream row)): on a shared column name the precedence is
// stream > steps[0] > steps[1] > ... (csvplus.go:559-560 nested).
func chainSource(spec *chainSpec) DataSource {
	return func(fn RowFunc) error {
		return batched(spec.upstream, spec.steps[0].columns, func(batch []Row) error {
			if len(batch) == 0 {
				return nil
			}
			if len(spec.steps) == 1 {
				return probeBatch(spec.steps[0].index, spec.steps[0].columns, false, batch, fn)
			}
			// origin[k]: 0 = step k's key columns come from the stream rows, t+1 = from the rows of steps[t].index (round 5:
			// cph_chain_step.source — people.Join(orders, "id").Join(products), csvplus_test.go:280-285, stays ONE device call)
			origin := make([]int, len(spec.steps))
			for k := 1; k < len(spec.steps); k++ {
				org := -2
				for _, col := range spec.steps[k].columns {
					o := columnOrigin(spec, k, col, batch)
					if o < 0 || (org != -2 && o != org) {
						return runSteps(spec, 0, batch, fn) // the rows disagree, or the column is missing: step by step
					}
					org = o
				}
				origin[k] = org
			}
			return fusedBatch(spec, origin, batch, fn)
		})
	}
}

// columnOrigin says where the value of `col` in the row that step k's Join sees comes from, for a whole batch: 0 = every
// stream row carries it (mergeRows, :571-583, lets the stream's value win), t+1 = no stream row does and every row of
// steps[t].index does (the earliest such t < k: the nested merges let the earlier, right-hand row win), -1 = the rows
// disagree or the column is missing (runSteps then reports what the reference reports).
func columnOrigin(spec *chainSpec, k int, col string, batch []Row) int {
	have := 0
	for _, r := range batch {
		if _, ok := r[col]; ok {
			have++
		}
	}
	if have == len(batch) {
		return 0
	}
	if have != 0 {
		return -1
	}
	// per (index, column) the answer is cached in the index's device twin (gpu.go columnPresence): the rows of an earlier index
	// are walked once per column, not once per batch
	gpuMu.Lock()
	defer gpuMu.Unlock()
	for t := 0; t < k; t++ {
		switch spec.steps[t].index.columnPresence(col) {
		case colAll:
			return t + 1
		case colSome: // SOME rows of an earlier index carry it: no single origin
			return -1
		}
	}
	return -1
}

// fusedBatch: one cph_join_chain_ex call, positions out.  origin[k] == 0: step k's key columns are staged from the stream rows;
// t+1: they are columns of steps[t].index's rows, staged once per index in sorted order (cph_chain_step.source = -(t+1)) — the
// device reads the key from the row that step matched (or answers it from build tables joined with each other first).
func fusedBatch(spec *chainSpec, origin []int, batch []Row, fn RowFunc) error {
	ctx, err := gpu()
	if err != nil {
		return err
	}
	gpuMu.Lock()
	stagePool.reset()
	steps := make([]C.cph_chain_step, len(spec.steps))
	keep := make([][]C.cph_strcol, len(spec.steps)) // the descriptors must outlive the call
	for k, st := range spec.steps {
		if origin[k] == 0 {
			cols, err := stagePool.stage(ctx, batch, st.columns)
			if err != nil {
				gpuMu.Unlock()
				return err
			}
			keep[k] = cols
			steps[k] = C.cph_chain_step{index: st.index.gpu.h, cols: &cols[0], ncols: C.int32_t(len(cols))}
			continue
		}
		from := spec.steps[origin[k]-1].index
		cols := make([]C.cph_strcol, len(st.columns))
		for c, name := range st.columns {
			sc, err := from.sideColumn(ctx, name)
			if err != nil || sc == nil { // (columnOrigin saw the column in every row: only an allocation can fail here)
				gpuMu.Unlock()
				if err == nil {
					err = errors.New("csvplus: build-side key column vanished")
				}
				return err
			}
			cols[c] = *sc
		}
		keep[k] = cols
		steps[k] = C.cph_chain_step{index: st.index.gpu.h, cols: &cols[0], ncols: C.int32_t(len(cols)), source: C.int32_t(-origin[k])}
	}
	var ch *C.cph_chain
	rc := C.cph_join_chain_ex(ctx, &steps[0], C.int32_t(len(steps)), 0, C.CPH_MEM_HOST, C.CPH_CHAIN_POSITIONS, &ch)
	if rc != C.CPH_OK {
		e := lastErr(ctx)
		gpuMu.Unlock()
		return e
	}
	gpuMu.Unlock()
	defer func() { gpuMu.Lock(); C.cph_chain_release(ch); gpuMu.Unlock() }()
	n := int(ch.nrows)
	var streamRow []uint64 // nil: every stream row joined exactly once, result row m is stream row m
	if ch.stream_row != nil {
		streamRow = unsafe.Slice((*uint64)(unsafe.Pointer(ch.stream_row)), n)
	}
	pos := make([][]uint32, len(steps))
	for k := range steps {
		pos[k] = unsafe.Slice((*uint32)(unsafe.Pointer(ch.build_row[k])), n)
	}
	for m := 0; m < n; m++ { // emission order: stream row, then position in steps[0]'s index, then in steps[1]'s, ...
		r := m
		if streamRow != nil {
			r = int(streamRow[m])
		}
		row := mergeRows(spec.steps[0].index.impl.rows[pos[0][m]], batch[r])
		for k := 1; k < len(steps); k++ {
			row = mergeRows(spec.steps[k].index.impl.rows[pos[k][m]], row)
		}
		if err := fn(row); err != nil {
			return err
		}
	}
	_ = keep
	return nil
}

// Except (:588-608).
func (src DataSource) Except(index *Index, columns ...string) DataSource {
	if len(columns) == 0 {
		columns = index.impl.columns
	} else if len(columns) > len(index.impl.columns) {
		panic("too many source columns in Except()")
	}
	return func(fn RowFunc) error {
		return batched(src, columns, func(batch []Row) error { return probeBatch(index, columns, true, batch, fn) })
	}
}
