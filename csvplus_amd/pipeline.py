"""The reference's README pipeline (README.md:33-66) with every stage on the device:

    customers.csv -> UniqueIndexOn(id)          orders.csv -> SelectColumns(...)
    products.csv  -> UniqueIndexOn(prod_id)        .Join(customers, "cust_id").Join(products, "prod_id").ToCsv(...)

CSV bytes -> columns (cph_csv_parse) -> indices (cph_index_build) -> fused chained join (cph_join_chain) ->
output columns (cph_gather_rows: mergeRows, csvplus.go:571-583, column by column) -> CSV bytes
(cph_csv_write: ToCsv, :379-406).  Nothing leaves HBM between the first and the last step.
"""
from __future__ import annotations

import ctypes as C
import time

from . import _native as N
from . import ingest
from .materialize import gather_rows


class Table:
    """A parsed CSV file: device columns by name."""

    def __init__(self, csv_table: ingest.CsvTable):
        self.t = csv_table
        self.cols = {nm.decode(): col for nm, col in zip(csv_table.names, csv_table.columns)}
        self.nrows = csv_table.nrecords

    def __getitem__(self, name):
        return self.cols[name]

    def release(self):
        self.t.release()


def read_table(ctx: N.Context, text: bytes, select=None, **kw) -> Table:
    t = ingest.read_csv(ctx, text, select=select, out_mem=N.CPH_MEM_DEVICE, **kw)
    if t.error_kind:
        err = ingest.CsvError(t.error_kind, t.error_record)
        t.release()
        raise err
    return Table(t)


def join_to_csv(ctx: N.Context, stream: Table, steps, out_columns, timings: dict | None = None, out_mem: int = N.CPH_MEM_HOST,
                fused: bool = True, positions: bool | None = None):
    """steps: [(index_table, index_key_column, stream_key_column), ...] — each index must be unique on its key
    (UniqueIndexOn; a duplicate raises like the reference's error :751).  out_columns: [(output name, table,
    column)] where table is `stream` or one of the index tables; the caller resolves name collisions the way
    mergeRows does (the stream's column wins, :578-580) by naming the table it wants.
    positions: the Join reports SORTED POSITIONS (cph_join_chain_ex CPH_CHAIN_POSITIONS — the
    reference's own row handle, csvplus.go:553-567, and the cheap lookup on the device) and the payload columns of every
    build table are put in index order once (cph_index_permute: the reference's createIndex leaves its rows sorted, :736),
    so a position IS the row subscript; False: original row ids into the tables as they were read (rounds 1-3); None
    (default): whichever is cheaper by the measured costs (profiles/r04_pipeline.txt) — putting a payload column in index
    order costs ~0.3 ms per 1e7 table rows, reporting positions saves ~1.0 ms per 1e8 stream rows, so a one-shot pipeline
    whose stream is only a few times longer than its build tables keeps row ids.
    Returns the CSV text (header + joined rows, stream order): bytes, or a DeviceBytes handle for out_mem DEVICE."""
    def lap(name, t0):
        if timings is not None:
            ctx.synchronize()
            timings[name] = timings.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

    if positions is None:
        payload = {}
        for _, tab, col in out_columns:
            if tab is not stream:
                payload[(id(tab), col)] = tab.nrows
        positions = 1.0e-8 * stream.nrows > 3.0e-8 * sum(payload.values())
    indices, ch, bufs = [], None, []
    try:   # whatever fails below, the indexes, the chain and the gathered columns go back to the ctx pool
        t0 = time.perf_counter()
        for tab, key, _ in steps:
            ix = N.DeviceIndex(ctx, [tab[key]], unique=True)
            indices.append(ix)
            if ix.status == N.CPH_ERR_DUPLICATE:
                raise ValueError(f"duplicate value while creating unique index on {key!r} (sorted position {ix.first_dup})")
        lap("index_ms", t0)
        t0 = time.perf_counter()
        tabs = [t for t, _, _ in steps]
        sorted_cols = {}   # (table number, column) -> that column in index order
        if positions:
            from .materialize import permute_col
            for _, tab, col in out_columns:
                if tab is not stream and (tabs.index(tab), col) not in sorted_cols:
                    cb = permute_col(ctx, indices[tabs.index(tab)], tab[col])
                    bufs.append(cb)
                    sorted_cols[(tabs.index(tab), col)] = cb.as_device_strcol()
        lap("index_ms", t0)
        t0 = time.perf_counter()
        ch = N.join_chain(ctx, [(ix, [stream[skey]]) for ix, (_, _, skey) in zip(indices, steps)], out_mem=N.CPH_MEM_DEVICE,
                          positions=positions)
        ptrs = ch.device_ptrs()
        n = ch.nrows
        lap("join_ms", t0)
        t0 = time.perf_counter()
        from .materialize import csv_write
        cols, ids = [], []
        for _, tab, col in out_columns:
            cols.append(tab[col] if tab is stream or not positions else sorted_cols[(tabs.index(tab), col)])
            if tab is stream:
                ids.append(None if ch.identity or n == 0 else (ptrs["stream_row"], 64, n))
            else:
                ids.append((ptrs["build_row"][tabs.index(tab)], 32, n))
        if n == 0:
            cols, ids = [c.head(0) for c in cols], [None] * len(cols)
        if fused:
            # mergeRows inside the writer: fields are read through the row-id tuples, nothing is materialised
            text = csv_write(ctx, cols, [name for name, _, _ in out_columns], out_mem=out_mem, row_ids=ids, nrows=n)
            lap("to_csv_ms", t0)
        else:
            gcols = []
            for c, i in zip(cols, ids):
                if i is None and c.nrows == n:
                    gcols.append(c)
                    continue
                cb = gather_rows(ctx, c, i, out_mem=N.CPH_MEM_DEVICE)
                bufs.append(cb)
                gcols.append(cb.as_device_strcol())
            lap("gather_ms", t0)
            t0 = time.perf_counter()
            text = csv_write(ctx, gcols, [name for name, _, _ in out_columns], out_mem=out_mem)
            lap("to_csv_ms", t0)
        return text
    finally:
        for cb in bufs:
            cb.release()
        if ch is not None:
            ch.release()
        for ix in indices:
            ix.close()
