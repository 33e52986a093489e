/*
 * csvplus_hip.h — C ABI of libcsvplus_hip: the MI355X (gfx950) implementation
 * of csvplus's Index-build + Join hot path.
 *
 * The reference (maxim2266/csvplus, one Go file) has no FFI layer; the seams a
 * cgo shim replaces are (all citations are file:line in the reference):
 *
 *   sort.Sort(&index.impl)                      csvplus.go:736   -> cph_index_build
 *   createUniqueIndex adjacent-equal scan       csvplus.go:749-753 -> cph_index_build(unique=1)
 *   Join per-row  first() + forward scan        csvplus.go:557-563 -> cph_join_probe
 *   Except per-row has()                        csvplus.go:599-602 -> cph_join_probe (cnt==0 rows)
 *   indexImpl.find (Find / SubIndex bounds)     csvplus.go:870-891 -> cph_index_find
 *
 * Conventions
 *   - Every function returns an int32 status (CPH_OK == 0, errors < 0) unless
 *     noted.  cph_last_error(ctx) gives a NUL-terminated message owned by the
 *     ctx, valid until the next call on that ctx.
 *   - Inputs are borrowed for the duration of the call only.  Outputs
 *     (cph_index, cph_matches) are library-owned and explicitly released.
 *   - A cph_ctx is single-threaded (the reference has no concurrency either:
 *     no go/sync/chan in csvplus.go).  The library calls hipSetDevice on every
 *     entry, so a ctx may migrate between OS threads (cgo does that).
 *   - No callbacks into the caller.  No exceptions cross the boundary.
 *   - String columns are Arrow-style: `data` holds the concatenated raw value
 *     bytes (no terminators, any byte value incl. NUL), `offsets` has
 *     nrows+1 monotonically non-decreasing entries of 32 or 64 bits.
 *     `mem` says where both arrays live: host memory (pageable or pinned) or
 *     device memory of the ctx's GPU.
 *   - Row ids are uint32: a single table is limited to 2^32-1 rows
 *     (CPH_ERR_TOO_MANY_ROWS otherwise).  Go `int` row counts are 64-bit; the
 *     shim must check.
 *
 * Ordering contract (csvplus.go:794-807, indexImpl.Less): rows are ordered by
 * the tuple of key columns, each compared with strings.Compare = unsigned
 * bytewise lexicographic, a proper prefix sorting first.  Rows with equal keys
 * keep their input order (stable), which is one of the orders the reference's
 * unstable sort.Sort may produce (parity class P1).
 */
#ifndef CSVPLUS_HIP_H
#define CSVPLUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPH_API __attribute__((visibility("default")))

/* ---- status codes ---------------------------------------------------------- */
enum {
    CPH_OK                 = 0,
    CPH_ERR_INVALID        = -1,  /* bad argument (NULL, nkeycols<=0, offset_bits...)        */
    CPH_ERR_HIP            = -2,  /* a HIP runtime call failed; see cph_last_error           */
    CPH_ERR_NO_DEVICE      = -3,  /* no usable GPU: the library never falls back to the CPU  */
    CPH_ERR_DUPLICATE      = -4,  /* unique index requested and equal keys exist             */
    CPH_ERR_TOO_MANY_ROWS  = -5,  /* nrows > 2^32-1                                          */
    /* -6 is retired: key length is unlimited (keys beyond 128 byte positions are cut into codec windows) */
    CPH_ERR_TOO_MANY_COLS  = -7,  /* join with more columns than the index has
                                     (csvplus.go:548-550 panics "too many source columns")   */
    CPH_ERR_NOMEM          = -8
};

#define CPH_MAX_KEY_COLS  16

enum { CPH_MEM_HOST = 0, CPH_MEM_DEVICE = 1 };

typedef struct cph_ctx   cph_ctx;
typedef struct cph_index cph_index;

/* One string column (or a row range of one). */
typedef struct {
    const uint8_t* data;        /* concatenated value bytes                               */
    const void*    offsets;     /* nrows+1 entries; value i = data[offsets[i]..offsets[i+1]) */
    uint64_t       nrows;
    int32_t        offset_bits; /* 32 or 64                                               */
    int32_t        mem;         /* CPH_MEM_HOST or CPH_MEM_DEVICE                         */
    uint32_t       fixed_width; /* > 0: every value has exactly this many bytes, value i =
                                   data[i*fixed_width ..); offsets is ignored (may be NULL).
                                   Saves the offsets' HBM traffic and a dependent load.   */
    uint32_t       reserved_;
} cph_strcol;

/* One key value (for cph_index_find). Host memory. */
typedef struct {
    const uint8_t* data;
    uint64_t       len;
} cph_strval;

/*
 * Result of a probe.  All arrays live in `mem` (as requested by the caller)
 * and stay valid until cph_matches_release.
 *
 *   lo[i], cnt[i]   for probe row i: the matching index rows are the sorted
 *                   positions [lo[i], lo[i]+cnt[i]).  cnt==0 means no match
 *                   (lo is then unspecified) — the reference's
 *                   `for i := first(values); i<n && !cmp(i,values,false); i++`
 *                   (csvplus.go:559) runs cnt[i] times.
 *   probe_idx[m], build_row[m]   m < nmatches: the joined pairs in the
 *                   reference's emission order (stream order, then ascending
 *                   index position, csvplus.go:553-567).  probe_idx is
 *                   probe_base + the row's position in this call; build_row is
 *                   the ORIGINAL row id of the index row (its position in the
 *                   table handed to cph_index_build), i.e. perm[lo+j].
 */
typedef struct {
    uint64_t        nprobe;
    uint64_t        nmatches;
    const uint32_t* lo;
    const uint32_t* cnt;
    const uint64_t* probe_idx;
    const uint32_t* build_row;
    int32_t         mem;
    int32_t         reserved_;
} cph_matches;

/* ---- context --------------------------------------------------------------- */

/* Creates a context bound to HIP device `device_id`.  Fails with
 * CPH_ERR_NO_DEVICE when the device does not exist (there is no CPU path). */
CPH_API int32_t cph_ctx_create(int32_t device_id, cph_ctx** out);
CPH_API void    cph_ctx_destroy(cph_ctx* ctx);
CPH_API const char* cph_last_error(const cph_ctx* ctx);

/* All work of this ctx is enqueued on `hip_stream` (a hipStream_t; NULL = the
 * ctx's own stream).  Lets a host framework (torch, a Go scheduler) order the
 * library's kernels with its own copies. */
CPH_API int32_t cph_ctx_set_stream(cph_ctx* ctx, void* hip_stream);

/* Measurement / tuning knobs (tools/microbench), per ctx — two devices in one process never share them:
 *   "chain_debug"   attribution switches of the chained-join kernel: bit 0 skips the table lookups, bit 1 the
 *                   key encode, bit 2 the result stores.  Results are WRONG while any bit is set; 0 = product path.
 *   "sort_threads"  256 / 512: workgroup size of the radix sort (0 = automatic)
 *   "sort_rbits"    8 / 9: digit width of the radix sort (0 = automatic)
 *   "sort_xcd_tiles" 0 / 1: the scatter kernel gives every XCD a contiguous range of tiles (default 1)
 *   "sort_digit_stream" 0 / 1 (default 1): a radix scatter over 64-bit keys leaves the next pass's digits behind as a byte stream, so that
 *                   pass's histogram reads 1 byte per key instead of 8
 *   "chain_nt_streams" 0 / 1 / 2 (default 0): the chained Join loads the stream's bytes and stores its results non-temporally — never,
 *                   always, in positions mode only (measured: within 1 %)
 *   "stream_role_streams" 0 / 1 (default 0): cph_stream_join in fused mode with one HIP stream for all uploads and one for all downloads
 *                   instead of one stream per slot (slower at 2 and 4 slots)
 *   "stream_zero_copy_out" 0 / 1 (default 0): cph_stream_join in fused mode lets the kernel store its results straight into the slot's
 *                   pinned host block
 *   "speculative_groups"  dictionary windows of long keys on large inputs: 0 = always the exact pass over all rows,
 *                   1 = dictionaries straight from the row sample when every value of every chosen window is COMMON in it
 *                   (met 16 times or more: a closed vocabulary; default),
 *                   2 = always from the sample (the encode kernel completes them; a test hook)
 *   "plan_threads" / "gstats_threads"  workgroup sizes of the dictionary encode / window statistics kernels (0 = default)
 *   "join_hash"     0: indexes built on this ctx never get a hash table — their sparse-key Joins binary-search the
 *                   sorted codes as before (A/B switch for measurements and for the fallback's tests; default 1)
 *   "codec_split"   0 / 1 (default 1): keys that do not code in 32 bits are tried with the delimiter split — the costliest
 *                   variable-length key column cut at its first delimiter byte into a dictionary-coded prefix and a
 *                   per-position suffix (floating fields like "Smith/Amelia#12345": 25 bits instead of 47; A/B switch)
 *   "direct_fused_encode" 0 / 1 (default 1): the direct sort of ONE fixed-width 8-byte key column under an arithmetic codec (decimal ids; the
 *                   column 16-byte aligned) codes the keys inside its first partition level — no encode kernel, no 4-byte code per row written
 *                   and read again (1e8 ids: 1.30 -> 1.12 ms; A/B switch)
 *   "direct_ranktab" 0 / 1 (default 1): the window sort of a code space larger than its table writes the Join's rank table (presence bits +
 *                   keys before, 8 bytes per 32 codes) while it streams its windows out: a Join that reports positions finds it ready
 *                   (0: built by the first such Join, k_build_ranktab; A/B switch)
 *   "chain_prejoin" 0 / 1 (default 1): chain steps whose key is a column of an earlier step's build table (cph_chain_step.source != 0)
 *                   are answered from PRE-JOINED tables when the stream is at least twice as long as those tables: the build sides are
 *                   joined with each other first (one pass over the table's column), the stream rows then need one 4-byte gather per
 *                   such step (0: the fused kernel gathers and encodes the key per stream row; A/B switch)
 *   "hash_load_pct" 25..90 (default 50): load factor of the Join hash tables (sparse key spaces), per cent of a 64-byte sector's slots,
 *                   counted in DISTINCT keys (rounds 3-4 counted rows).  Measured at 1e7 keys x 1e8 probes: 50 % 2.69 ms, 75 % 3.58 ms,
 *                   85 % 5.5 ms — a smaller table does not pay for the longer probe sequences (profiles/r05_tried_not_kept.txt)
 *   "split_speculative" 0 / 1 (default 1): IndexOn over ONE variable-length key column of >= 2^22 rows takes the split codec's prefix
 *                   dictionary and suffix alphabets from its 2^18-row sample alone; the encode kernel checks every row against them
 *                   (prefix in the dictionary, every suffix byte and the end of the suffix in its position's alphabet, lengths within
 *                   the sample's) and a row it cannot code makes the build start over with the exact statistics pass — the index
 *                   is the same either way (A/B switch: 0 = always the exact pass)
 *   "scan_lookback" 0 / 1 (default 1): the 32-bit exclusive scans (radix count matrices, compaction offsets) run as ONE launch
 *                   (decoupled look-back) instead of three (A/B switch)
 *   "counted_sort"  0 / 1 (default 1): IndexOn over one-word 32-bit codes WITH duplicates (>= 2^21 rows) sorts through counted LDS windows
 *                   (rows per window of the code space counted first, two partition levels into exactly sized buckets, every window
 *                   counting-sorted in LDS with equal keys put in row order) instead of 3-4 radix passes; a window beyond 16384 rows
 *                   (one key with more duplicates, a dense cluster in a sparse code space) is noticed on the device and the classic
 *                   passes sort the same codes.  Same index either way (A/B switch)
 *   "csv_fast"      0 / 1 (default 1): cph_csv_parse first tries byte-parallel passes over 16 KiB text tiles (texts without any quote,
 *                   no TrimLeadingSpace, <= 8 columns, < 4 GiB, no blank / comment line inside, records that end within 4 KiB of their
 *                   tile); anything else — and every error — goes through the record-parallel kernels.  Same columns either way
 *   "csv_onepass"   0 / 1 / N (default 1): cph_csv_write[_rows] over >= 4096 rows with at least one such group renders the text in ONE pass: every
 *                   group of adjacent columns gathered from one table (<= the output's rows) through one row-id array is rendered once per
 *                   table row into a slot table ([length][CSV text], stride 16..128 bytes: one aligned fetch per output row and table),
 *                   a tile of 256 / 512 records learns its place in the text from a decoupled look-back over the tiles before it
 *                   (persistent grid, no length array, no scan) and leaves through LDS.  Not taken (-> the two-pass writer, same bytes):
 *                   more than 8 output columns after grouping, a fragment beyond 127 bytes, records beyond ~72 bytes on average, a
 *                   buffer estimate (stream columns' bytes + 1/8 for quotes) that a tile overruns.  0: always the two passes; N > 1: the
 *                   one pass whatever the row count, with at most N workgroups (tests)
 *                   The one-pass kernel is a persistent grid whose tiles wait for the tiles in front of them: the library runs one at a time per
 *                   device and process; two PROCESSES that write CSV on the same GPU at the same moment should set 0
 *   "csv_onepass_debug" bits (default 0; measurement only, the text is wrong when set): 1 = no look-back, 2 = no record bytes, 4 = one
 *                   record per thread
 *   "direct_sort"   0 / 1 (default 1): a build that expects distinct keys (cph_index_build with unique = 1, cph_index_spec.unique) over a
 *                   dense 32-bit code space (rows <= code states <= 2 rows: decimal ids, row numbers) sorts without radix passes: the rows
 *                   are split by the top bits of their codes into 2^14-slot windows, every window is filled in LDS (slot = code) and
 *                   streamed out — all global stores sequential; a duplicate is noticed on the device and the build starts over the
 *                   general way, which also reports where the first duplicate is.  (A/B switches: 4 = slot[code] = row as one random
 *                   4-byte store per row, the round-4 path; 2 = that behind one radix pass on the top 8 bits; 3 = the encode kernel
 *                   fills the slots of a full code space itself)
 *   "stats_sample"  0 / 1 (default 1): IndexOn over ONE fixed-width key column (<= 40 bytes) of >= 2^20 rows learns its per-position
 *                   alphabets from ~65 536 rows spread over the table instead of a pass over all rows; the encode kernel checks every
 *                   row against them, and a row with a byte the sample did not show makes the build start over with the exact
 *                   pass (the index is the same either way; A/B switch)
 *   "host_build"    0 / 1 (default 1): cph_index_build over ONE key column of at most 8 byte positions (decimal ids, short tags) of
 *                   >= 2^20 rows that lives in HOST memory forms the key codes on the host — alphabets from a sample of the rows,
 *                   the codes by the ctx's worker pool in 2^22-row chunks, each uploaded (4 bytes per row instead of the strings)
 *                   while the next is coded — and the device only sorts; a row the sampled alphabets cannot code, or duplicates under
 *                   the optimistic direct sort, send the build down the general path (upload of the strings).  Same index either way.
 *   "host_threads"  threads of that worker pool (default 0: half the hardware threads, at most 32: the loops are bound by the
 *                   memory bandwidth of the NUMA node that holds the pinned buffers well before that)
 *   "host_split"    0 / 1 / 2 (default 1): the same for ONE variable-length key column of >= 2^22 rows in HOST memory whose values
 *                   want the delimiter split ("Smith/Amelia#12345": a prefix dictionary + suffix positions, 25 bits for BASELINE
 *                   config 3's 10-22 byte keys): the codec comes from the sample the device would take, read by the host's threads,
 *                   the codes (4 bytes per row instead of ~22 bytes of string) from a loop that checks every row like the device's
 *                   encode kernel — unknown prefix, foreign suffix byte, longer value: the strings are uploaded after all.  Same index.
 *                   1: taken when the estimate says the host's threads beat the upload (~20 ns per row and thread against ~50 GB/s of
 *                   PCIe; the threads counted are those the cgroup's CPU QUOTA allows — a container throttled to 16 CPUs codes 1e8
 *                   rows in ~100 ms, the upload takes 44), 2: always, 0: never
 *   "host_split_threads" threads of that loop's own pool (default 0: 3/8 of the hardware threads, at most 96, at most the CPU quota —
 *                   it computes, where the 8-byte loops wait for memory)
 *   "host_numa"     0 / 1 (default 0): 1 binds that pool's workers to the CPUs of the NUMA node that holds the column's first page
 *                   (opt-in: its effect could not be measured on the quota-limited test hosts)
 *   "sample_lean"   0 / 1 (default 1): the sample of a fixed-width key column of at most 8 bytes is taken by a kernel that collects
 *                   nothing but the per-position byte presence (the only thing stats_sample uses; A/B switch)
 *   "hash_partitioned" 0 / 1 / 2 (default 1): the hash table of a duplicate-free index of >= 2^21 keys (sparse key codes: random ids,
 *                   hashes) is built SLICE BY SLICE: the rows grouped by the 64 KB slice of the table their home sector lies in, every
 *                   slice filled in LDS by one workgroup and written out once (probe sequences wrap inside a slice) instead of
 *                   compare-and-swaps all over the table; 0: the latter; 2: slice by slice whatever the size (tests).  Lookups answer
 *                   the same either way
 *   "build_side_stream" 0 / 1 (default 1): cph_index_build_many enqueues every second build of a batch on a second stream of
 *                   the ctx, so the launch-latency-bound kernels of a small table run beside its neighbour's instead of behind
 *                   them; both streams are idle when the call returns (A/B switch)
 *   "codec_debug"   1: the window choice of every index build (and the phase times of a one-launch build) go to stderr
 *   "small_build_rows"  tables of at most this many rows (default 8192, at most 16384) are indexed by ONE launch of one
 *                   workgroup and one synchronisation (small_build.hip); 0 = always the general path
 *   "probe_hash_rows"  2 / 4: rows per phase of the generic hash probe (default 2)
 *   "chain_rank_lds"  0 / 1: a Join that reports positions copies the rank tables of small indexes into LDS (default 1)
 *   "chain_rows4"   0 / 1 / 2 (default 1): the register-heavy variants of the fused chained Join (keys beyond 8 bytes with 64-bit codes
 *                   from two steps on, long or wide chains from three steps on) keep 4 rows per lane in flight instead of 8, which
 *                   lets three waves per SIMD run instead of one; 0 = always 8, 2 = every general (non-lean) chain 4 (A/B switch)
 *   "chain_arith"   0 / 1 (default 1): the fused chained Join encodes fixed-width key columns over contiguous alphabets
 *                   (decimal ids) arithmetically — one aligned 8-byte load and a dot product instead of a LUT walk (A/B switch)
 *   "chain_identity" 0 / 1 (default 1): reporting positions, an index whose code space has exactly as many states as the
 *                   index has rows needs no lookup: the position of a key is its code (A/B switch)
 *   "pool_reserve_mb"  reserves ONE device slab of that many MiB now; later requests are carved out of it first
 *                   (first fit, coalesced on release) and only fall back to hipMalloc when it cannot serve them —
 *                   a one-shot caller pays its device allocations here, not inside its first call
 *   "pool_guard"    1: every device block the ctx allocates from now on carries 256 canary bytes behind the bytes
 *                   its user asked for, verified (after a device synchronisation) when the block is released
 *   "pool_guard_check"  verifies the canaries of all live blocks now; CPH_ERR_INVALID + the first violation's
 *                   description if any canary was ever overwritten
 * Unknown names fail with CPH_ERR_INVALID. */
CPH_API int32_t cph_ctx_set_option(cph_ctx* ctx, const char* name, int64_t value);

/* Blocks until everything enqueued by this ctx has finished. */
CPH_API int32_t cph_ctx_synchronize(cph_ctx* ctx);

/* Pinned (page-locked) host staging memory: cgo must not hand Go-heap pointers
 * that C retains, and pinned buffers make H2D/D2H copies asynchronous.
 * Page-locking costs ~90 us per block — more than a Join of a batch of 8192 rows (60-80 us) — so blocks of at most 4 MB
 * handed back through cph_pinned_free are kept by the ctx (up to 32) and handed out again; a host side may allocate and
 * free its staging per batch (round 5; before, only a caller that kept its blocks — the Go shim's stagePool — avoided it).
 * A block must not be freed while a call that reads or writes it is still in flight (cph_stream_join_submit ... next). */
CPH_API int32_t cph_pinned_alloc(cph_ctx* ctx, size_t bytes, void** out);
CPH_API int32_t cph_pinned_free(cph_ctx* ctx, void* p);

/* ---- index build: IndexOn / UniqueIndexOn (csvplus.go:529-537, 707-756) ----- */

/*
 * Builds the sorted index over `nkeycols` key columns (all with the same
 * nrows), leftmost column most significant.
 *
 *   unique != 0  mirrors createUniqueIndex: if two rows have equal keys the
 *                call returns CPH_ERR_DUPLICATE and *first_dup_pos is the
 *                smallest sorted position i>=1 with rows[i-1]==rows[i] on the
 *                key columns (the pair csvplus.go:749-753 reports; the shim
 *                formats rows[perm[i]] into the error text of :751).  The index
 *                is still returned so the shim can read perm[i]; it must destroy
 *                it to mirror the reference's nil return.
 *   unique == 0  *first_dup_pos is still filled (UINT64_MAX when all keys are
 *                distinct); duplicates are not an error.
 *
 * The index stays resident on the GPU (sorted key codes + permutation).
 */
CPH_API int32_t cph_index_build(cph_ctx* ctx, const cph_strcol* keycols, int32_t nkeycols, int32_t unique,
                                cph_index** out, uint64_t* first_dup_pos);
/*
 * Several IndexOn calls as one batch (the two build sides of orders.Join(customers).Join(products),
 * README.md:56): the same results as nspecs calls of cph_index_build, but the batch pays the build's two
 * host round trips (key statistics, first duplicate) ONCE instead of once per index, and the kernels of the
 * different indexes queue back to back.  out[i] / first_dup_pos[i] / status[i] are per index (status and
 * first_dup_pos may be NULL); the return value is the first non-OK status (CPH_ERR_DUPLICATE for a `unique`
 * spec with equal keys — that index is still returned, as by cph_index_build).
 */
typedef struct cph_index_spec {
    const cph_strcol* keycols;
    int32_t           nkeycols;
    int32_t           unique;
} cph_index_spec;
CPH_API int32_t cph_index_build_many(cph_ctx* ctx, const cph_index_spec* specs, int32_t nspecs, cph_index** out,
                                     uint64_t* first_dup_pos, int32_t* status);
CPH_API void    cph_index_destroy(cph_index* index);

/* Number of rows / key columns of the index. */
CPH_API uint64_t cph_index_nrows(const cph_index* index);
CPH_API int32_t  cph_index_nkeycols(const cph_index* index);

/* perm[i] = original row id of the row at sorted position i.
 * mem = CPH_MEM_HOST: copied to library-owned pinned memory on first use;
 * mem = CPH_MEM_DEVICE: the resident device array.  Valid until destroy. */
CPH_API int32_t cph_index_perm(cph_index* index, int32_t mem, const uint32_t** perm, uint64_t* n);

/* ---- probe: Join / Except (csvplus.go:545-608) ------------------------------ */

/*
 * Probes `nprobecols` columns (1 <= nprobecols <= index key columns; fewer =
 * prefix join on the leading index columns, matched positionally,
 * csvplus.go:546-550, :910) against the index.
 *
 *   row_sel     optional (may be NULL): nsel row numbers into the probe columns,
 *               as uint32 (sel_bits = 32) or uint64 (sel_bits = 64) values from
 *               which sel_base is subtracted; probe row i is then row
 *               row_sel[i] - sel_base of the columns and nprobe = nsel.  Same
 *               memory space as the columns.  This is how a chained Join
 *               (README.md:56) stays on the device: the second probe passes the
 *               first join's probe_idx array (sel_bits = 64, sel_base = the first
 *               call's probe_base) and so probes exactly the rows the first join
 *               emitted, in emission order.
 *   probe_base  added to the local probe position in probe_idx (global row
 *               number of this chunk / shard's first row).
 *   want_pairs  0: only lo/cnt/nmatches are produced (Except, counting);
 *               1: probe_idx/build_row are expanded too.
 *   out_mem     where the result arrays should live.
 */
CPH_API int32_t cph_join_probe(cph_ctx* ctx, const cph_index* index, const cph_strcol* probecols,
                               int32_t nprobecols, const void* row_sel, int32_t sel_bits, uint64_t sel_base,
                               uint64_t nsel, uint64_t probe_base, int32_t want_pairs, int32_t out_mem,
                               cph_matches** out);
CPH_API void    cph_matches_release(cph_matches* m);

/* ---- chained Join on the device (README.md:56; csvplus.go:545-569 nested) ---- */

/* Joins per device call.  Chains of up to 4 Joins over duplicate-free single-column indexes run as ONE fused pass over the stream
 * rows; longer ones (and chains with duplicate keys / several key columns) run the general chain — probe, select, compose, one
 * step at a time — on the device, still inside one call (round 5: was 4, a fifth Join started a second call). */
#define CPH_MAX_CHAIN 8

/* One Join of a chain: stream.Join(index, cols...).  ncols <= the index's key columns.
 *   source == 0   `cols` are columns of the STREAM table (all such steps see the same nrows): the value the Join reads when
 *                 the stream row carries the column (mergeRows lets the stream's value win, csvplus.go:571-583).
 *   source == k   (1 <= k <= this step's number) `cols` are columns of the BUILD TABLE of step k-1 — the table steps[k-1].index
 *                 was built from, in that table's ORIGINAL row order (cols[j].nrows == cph_index_nrows(steps[k-1].index)): the
 *                 key of this Join is read from the row that step k-1 matched.  This is the reference's flagship chain,
 *                 people.Join(orders, "id").Join(products) (csvplus_test.go:280-285): prod_id is a column of the ORDERS index
 *                 row, which mergeRows copied into the joined row the second Join sees.
 *   source == -k  the same with `cols` laid out in step k-1's SORTED order (row i of cols = the row at sorted position i:
 *                 how the reference holds index.impl.rows after createIndex, csvplus.go:736; what cph_index_permute produces);
 *                 needs CPH_CHAIN_POSITIONS.
 * Step 0 always reads the stream. */
typedef struct {
    const cph_index*  index;
    const cph_strcol* cols;
    int32_t           ncols;
    int32_t           source;
} cph_chain_step;

/*
 * Result of stream.Join(i0, c0).Join(i1, c1)...: the joined rows as row-id tuples
 * in the reference's emission order (stream order; for one stream row the
 * matches of step 0 ascending by index position, and for each of them the
 * matches of step 1, ...).  stream_row[m] = probe_base + the stream row
 * (stream_row == NULL means the identity: every stream row joined exactly once, so
 * row m of the result is stream row probe_base + m and no array is materialised),
 * build_row[k][m] = ORIGINAL row id of the matching row of step k's index.
 * The host materialises row m as mergeRows(...mergeRows(index_k row, ...), stream row)
 * (csvplus.go:571-583).  Arrays live in `mem`, valid until cph_chain_release.
 */
typedef struct {
    uint64_t        nrows;
    const uint64_t* stream_row;
    const uint32_t* build_row[CPH_MAX_CHAIN];
    int32_t         nsteps;
    int32_t         mem;
    int32_t         positions;      /* 1: build_row[k][m] is the SORTED POSITION of the row in index k (see below) */
    int32_t         reserved_;
} cph_chain;

/*
 * A key of a later step is either a column of the stream table (source == 0) or a column of an earlier step's build table
 * (source != 0, see cph_chain_step) — between them the two cover every value mergeRows can hand to a later Join.  When every
 * index has distinct keys over a single column the whole chain runs as ONE pass over the stream rows, whatever the sources
 * (a build-side key is gathered from the matched row inside the kernel); otherwise the steps run one after the other ON THE
 * DEVICE (probe, expand, compose), still one call.
 */
CPH_API int32_t cph_join_chain(cph_ctx* ctx, const cph_chain_step* steps, int32_t nsteps, uint64_t probe_base,
                               int32_t out_mem, cph_chain** out);

/*
 * The same with flags.  CPH_CHAIN_POSITIONS: build_row[k][m] is not the original row id but the SORTED POSITION of the
 * matching row in index k — the value the reference itself works with: after createIndex the rows of an Index ARE in
 * sorted order (csvplus.go:736) and Join reads index.impl.rows[first() + i] (csvplus.go:553-567), so a host that keeps
 * its index rows sorted (as the reference does) reaches the joined row with one array access; the original row id is
 * cph_index_perm(index)[position].  For the device this removes the one random access per probe row that cannot be
 * cached: a duplicate-free index over a dense code space maps code -> position through presence bits and a running count
 * per 32 codes (8 bytes per 32 codes: 2.5 MB for a 1e7-code space, resident in every XCD's L2) instead of a 4-byte
 * row id per code (40 MB: one 64-byte Infinity-Fabric sector per probe row).
 */
#define CPH_CHAIN_POSITIONS 1u
CPH_API int32_t cph_join_chain_ex(cph_ctx* ctx, const cph_chain_step* steps, int32_t nsteps, uint64_t probe_base,
                                  int32_t out_mem, uint32_t flags, cph_chain** out);
CPH_API void    cph_chain_release(cph_chain* chain);

/* ---- multi-GPU: the exchange step of the row-range sharded Join ----- */

/*
 * One process (rank) per GPU joins a contiguous range of the stream rows against replicated build
 * sides; concatenating the per-rank row-id lists in rank order gives the reference's emission order
 * (stream order, csvplus.go:553-567).  A cph_dist wraps the communicator that moves those lists:
 * RCCL over xGMI (librccl.so is resolved on first use — the library does not link against it; a copy the host
 * process already loaded, e.g. torch's, is the one it binds to).
 *   rank 0:  cph_dist_unique_id(ctx, id)  -> ship the CPH_DIST_ID_BYTES bytes to every rank by any side
 *            channel the host program has (a Go host: its own RPC; torch: broadcast_object_list)
 *   all:     cph_dist_create(ctx, id, rank, nranks, &d)            (collective: ncclCommInitRank)
 * Every call below is collective: all ranks call it, in the same order, each on its own ctx.
 * Work is enqueued on the ctx's stream; the only host wait is for 24 bytes of counts per rank.
 */
#define CPH_DIST_ID_BYTES 128
#define CPH_MAX_GATHER    8
typedef struct cph_dist cph_dist;

CPH_API int32_t cph_dist_unique_id(cph_ctx* ctx, uint8_t* id /* CPH_DIST_ID_BYTES */);
CPH_API int32_t cph_dist_create(cph_ctx* ctx, const uint8_t* id, int32_t rank, int32_t nranks, cph_dist** out);
/* Test transport: the `nranks` ranks of `group` are THREADS of this process (one ctx each) sharing a GPU;
 * the collectives rendezvous in host memory and copy device to device.  Same code path above the transport. */
CPH_API int32_t cph_dist_create_loopback(cph_ctx* ctx, const char* group, int32_t rank, int32_t nranks, cph_dist** out);
CPH_API void    cph_dist_destroy(cph_dist* d);
CPH_API int32_t cph_dist_rank(const cph_dist* d);
CPH_API int32_t cph_dist_size(const cph_dist* d);
/* What moves the bytes, for logs and benchmark records: "rccl nranks=8 lib=<path> (the copy the host process had
 * loaded)" — the library binds to a librccl the process has ALREADY mapped (torch ships one) before it opens its
 * own, so one process never runs two RCCL copies — or "loopback nranks=...".  Owned by `d`, valid until the next call.
 * Environment: CPH_RCCL_LIBRARY=<path> makes the library bind to exactly that file (a particular RCCL build). */
CPH_API const char* cph_dist_transport(cph_dist* d);

/* Result of an allgatherv: data[a] (device memory of this rank's ctx, library-owned) holds `total`
 * elements of array a — rank 0's, then rank 1's, ...; counts / displs (host) say where each rank's begin.
 * Valid in stream order on the ctx's stream (cph_ctx_synchronize before reading from the host). */
typedef struct {
    uint64_t        total;
    int32_t         narrays, nranks;
    const uint64_t* counts;
    const uint64_t* displs;
    void*           data[CPH_MAX_GATHER];
    int32_t         mem;            /* CPH_MEM_DEVICE, or CPH_MEM_HOST: the arrays lie in the communicator's shared host
                                       buffer (cph_dist_join_chain with CPH_DIST_HOST_GATHER) */
    int32_t         reserved_;
} cph_gathered;

/* allgatherv of `narrays` device arrays that all hold `count` elements on this rank (elem_bytes[a] each):
 * ONE count exchange and ONE grouped send/recv batch for all arrays. */
CPH_API int32_t cph_dist_allgatherv(cph_dist* d, const void* const* send, const int32_t* elem_bytes, int32_t narrays,
                                    uint64_t count, cph_gathered** out);
CPH_API void    cph_gathered_release(cph_gathered* g);

/* The same for a chain result in device memory (cph_join_chain(..., probe_base, CPH_MEM_DEVICE)): gathers
 * build_row[0..nsteps) — data[0..nsteps) when *identity == 1: every rank's rows are its whole contiguous range,
 * so gathered row m is stream row *stream_base + m — or stream_row first (data[0], uint64) and the build rows
 * behind it (data[1..nsteps]) when some stream row of some rank did not join. */
CPH_API int32_t cph_dist_chain_allgather(cph_dist* d, const cph_chain* chain, uint64_t probe_base, cph_gathered** out,
                                         int32_t* identity, uint64_t* stream_base);

/*
 * The sharded chained Join in ONE call, its exchange pipelined behind the compute (round 4; csvplus.go:553-567 — rows are
 * independent, so a shard may be cut anywhere).  `steps` describe THIS rank's shard of the stream (its rows are stream rows
 * [probe_base, probe_base + nrows)); every rank passes the same chain.  The shard is cut into `nchunks` sub-chunks (0: the
 * library picks — one chunk per 2^24 rows of the largest shard, 1..8, and ONE for a single-rank communicator that keeps the result
 * on the device; the same value on every rank) and chunk k's rows travel while chunk k+1 is being joined:
 *   - default: to every peer over xGMI (RCCL send/recv batches on a second stream), straight into their final place in
 *     the gathered arrays — every rank ends up with the whole list in device memory, as with cph_dist_chain_allgather;
 *   - CPH_DIST_HOST_GATHER: each rank copies its chunks device -> host into ITS range of one host buffer that all ranks
 *     of the node share (POSIX shared memory, registered with HIP; created on first use and kept): no xGMI traffic at all,
 *     N PCIe links in parallel, and the joined list ends where a host program consumes it.  out->mem == CPH_MEM_HOST; the
 *     arrays stay valid until THIS rank's next CPH_DIST_HOST_GATHER call on the communicator or cph_dist_destroy (results
 *     alternate between two halves of the buffer, so a faster rank's next call does not touch what this rank still reads).
 * The pipeline needs results of a known size: a chain of duplicate-free single-column indexes yields at most one tuple per
 * stream row, so chunks travel in the dense form (slot == stream row) and the match totals are exchanged once at the end —
 * the call's only host wait; when some row of some rank did not join, the slots are compacted afterwards (one local pass).
 * Any other chain is joined whole and exchanged like cph_dist_chain_allgather does (stats->chunks == 0).
 *   shard_rows   optional: the row counts of ALL ranks' shards (nranks entries) when the caller knows them — as it does for
 *                a range split — and the ranges follow each other in rank order; NULL: one more count exchange up front.
 *   flags        CPH_CHAIN_POSITIONS (see cph_join_chain_ex) | CPH_DIST_HOST_GATHER | CPH_DIST_PACKED
 *   identity / stream_base / out: as cph_dist_chain_allgather.
 */
#define CPH_DIST_HOST_GATHER 0x100u
/* (round 6) xGMI mode only: the chunks travel BIT-PACKED — ceil(log2(index rows)) bits per step and row instead of 32 (the benchmark's
 * chain: 24 + 17 = 41 bits instead of 64), packed by the sender behind every sub-chunk's join, unpacked by the receiver behind the
 * exchange; the gathered arrays are the same.  Ignored (the plain format travels) when a row needs more than 64 bits, with
 * CPH_DIST_HOST_GATHER and for a single rank.  stats->packed_bits says what travelled. */
#define CPH_DIST_PACKED 0x200u
typedef struct {
    int32_t  chunks;               /* sub-chunks per shard; 0: the one-shot path ran                                      */
    int32_t  pipelined;            /* 1: more than one chunk, exchange and compute on separate streams                    */
    double   compute_ms;           /* ctx stream: first chunk's join enqueued -> last chunk's join done                   */
    double   exchange_ms;          /* exchange stream: first chunk ready -> match totals of all ranks received            */
    double   exposed_exchange_ms;  /* last chunk's join done -> totals received: the part of the exchange nothing hides   */
    double   total_ms;             /* first chunk's join enqueued -> totals received                                      */
    uint64_t bytes_sent, bytes_received;   /* this rank, row arrays only (xGMI; host gather: bytes_sent = PCIe bytes)     */
    int32_t  packed_bits;          /* CPH_DIST_PACKED in effect: bits per row on the wire; 0: plain 32 bits per step        */
    int32_t  reserved_;
} cph_dist_join_stats;
CPH_API int32_t cph_dist_join_chain(cph_dist* d, const cph_chain_step* steps, int32_t nsteps, uint64_t probe_base,
                                    const uint64_t* shard_rows, int32_t nchunks, uint32_t flags, cph_gathered** out,
                                    int32_t* identity, uint64_t* stream_base, cph_dist_join_stats* stats /* may be NULL */);

/* Build side, option B: rank `root` built `root_index` (others pass NULL); every other rank
 * receives an equal index (*out; NULL on the root) — descriptor, sorted codes and perm travel by ncclBroadcast
 * (12 bytes per row for one-word 64-bit codes) instead of every rank sorting the same table. */
CPH_API int32_t cph_dist_index_broadcast(cph_dist* d, const cph_index* root_index, int32_t root, cph_index** out);

/* ---- streaming Join of a host-resident stream (BASELINE config 5) -------------- */

/*
 * The probe side of a Join is a stream (csvplus.go:545-569 never materialises it).
 * When it lives in host memory it is fed chunk by chunk through a pipeline of
 * `nslots` slots, each with its own HIP stream, so that consecutive chunks overlap
 * upload, compute and download.
 *
 * cph_stream_join_create — chains the fused kernel accepts (every index has distinct
 * keys over ONE key column; CPH_ERR_INVALID otherwise): a submitted chunk's H2D copy,
 * kernels and D2H copy are enqueued without any host synchronisation.  Results are
 * DENSE per chunk (cph_stream_chunk.dense = 1): row r of the chunk joined iff bit
 * (r % 64) of match_bitmap[r / 64] is set, and then build_row[k][r] is the original
 * row id in index k.  Emission order is row order.  nmatches = number of set bits.
 *
 * cph_stream_join_create_general — ANY chain cph_join_chain takes (duplicate keys on
 * the build side, several key columns per step, prefix joins, keys of any length):
 * ncols[k] stream columns are step k's key, submit() takes the steps' columns one
 * after the other (ncols[0] + ncols[1] + ... columns).  The size of such a result is
 * only known after the probe, so every slot also owns a worker thread that runs the
 * general chain on the slot's stream; the pipeline overlaps across slots as before.
 * Results are PAIR LISTS per chunk (dense = 0), exactly cph_chain's layout:
 * nmatches joined rows in the reference's emission order, stream_row[m] (NULL: the
 * identity probe_base + m) and build_row[k][m]; match_bitmap is NULL.  A chain the
 * fused kernel accepts runs in the dense mode here as well.
 */
typedef struct cph_stream_join cph_stream_join;

typedef struct {
    uint64_t        probe_base;     /* as passed to submit                              */
    uint64_t        nrows;          /* rows of the chunk                                */
    uint64_t        nmatches;
    const uint64_t* match_bitmap;   /* ceil(nrows/1024)*16 words (>= nrows bits)        */
    const uint32_t* build_row[CPH_MAX_CHAIN];
    int32_t         nsteps;
    int32_t         dense;          /* 1: bitmap + one build row per chunk row; 0: pair lists of nmatches rows */
    const uint64_t* stream_row;     /* dense = 0: probe_base + chunk row of every joined row (NULL: identity)  */
    int32_t         positions;      /* 1: build_row[k] holds sorted positions in index k (cph_stream_join_set_positions) */
    int32_t         reserved_;
} cph_stream_chunk;

CPH_API int32_t cph_stream_join_create(cph_ctx* ctx, const cph_index* const* indexes, int32_t nsteps, int32_t nslots,
                                       cph_stream_join** out);
CPH_API int32_t cph_stream_join_create_general(cph_ctx* ctx, const cph_index* const* indexes, const int32_t* ncols,
                                               int32_t nsteps, int32_t nslots, cph_stream_join** out);
/* on != 0: every later chunk reports SORTED POSITIONS in build_row[k] instead of original row ids (CPH_CHAIN_POSITIONS,
 * see cph_join_chain_ex).  Call before the first submit. */
CPH_API int32_t cph_stream_join_set_positions(cph_stream_join* sj, int32_t on);
CPH_API void    cph_stream_join_destroy(cph_stream_join* sj);
/* step_cols[k] = the chunk's key column for step k: HOST memory (pinned — cph_pinned_alloc —
 * for real overlap), borrowed until the chunk has been returned by cph_stream_join_next.
 * Fails with CPH_ERR_INVALID when all slots are in flight. */
CPH_API int32_t cph_stream_join_submit(cph_stream_join* sj, const cph_strcol* step_cols, uint64_t probe_base);

/*
 * Key codes formed on the host (round 4): a stream in host memory then crosses PCIe as 4 bytes per row and step instead of
 * its key strings (17 bytes per row for the benchmark's two keys).  A stream row's key takes part in a Join only through
 * its code under the index's key codec, so for an index whose keys code in ONE word below 2^31 (cph_index_info:
 * code_words == 1, code_bits <= 31, dict_entries == 0, split == 0 — decimal ids, short tags) the host can form the code
 * with the table the device walks:
 *   cph_host_encoder_create   an encoder for `index` with a pool of `nthreads` worker threads (0: one per hardware
 *                             thread); CPH_ERR_INVALID when the index's codes do not qualify (ship the strings then)
 *   cph_host_encoder_run      cols = ALL key columns of the index for the chunk's rows (host memory) -> out_codes[nrows];
 *                             a key that cannot occur in the index gets CPH_CODE_ABSENT and joins nothing.  Blocks.
 *   cph_stream_join_submit_codes   like cph_stream_join_submit with step_codes[k] = the chunk's codes for step k (host
 *                             memory, pinned for real overlap; borrowed until the chunk was returned).  Fused-mode stream
 *                             joins only (cph_stream_join_create), every index with a direct lookup structure (dense code
 *                             space: cph_index_info.direct_table); results as for cph_stream_join_submit.
 * An encoder may be used from one thread at a time; it does not touch the GPU.
 */
#define CPH_CODE_ABSENT 0xFFFFFFFFu
typedef struct cph_host_encoder cph_host_encoder;
CPH_API int32_t cph_host_encoder_create(const cph_index* index, int32_t nthreads, cph_host_encoder** out);
CPH_API int32_t cph_host_encoder_threads(const cph_host_encoder* enc);
CPH_API int32_t cph_host_encoder_run(cph_host_encoder* enc, const cph_strcol* cols, int32_t ncols, uint32_t* out_codes);
CPH_API void    cph_host_encoder_destroy(cph_host_encoder* enc);
CPH_API int32_t cph_stream_join_submit_codes(cph_stream_join* sj, const uint32_t* const* step_codes, uint64_t nrows, uint64_t probe_base);
CPH_API int32_t cph_stream_join_pending(const cph_stream_join* sj);
/* Waits for the OLDEST chunk in flight.  The arrays are pinned memory owned by the
 * pipeline.  Slots are used round robin — chunk number k (counting submissions from 0)
 * lives in slot k % nslots — so the arrays of chunk k stay valid until chunk k + nslots
 * is SUBMITTED: a caller that keeps at most nslots - 1 chunks in flight can read a
 * returned chunk while the following ones are being processed. */
CPH_API int32_t cph_stream_join_next(cph_stream_join* sj, cph_stream_chunk* out);

/* ---- materialisation: the step after the path ------------------ */

/* A library-owned string column (64-bit offsets) living in col.mem. */
typedef struct {
    cph_strcol col;
    uint64_t   nbytes;      /* total value bytes */
} cph_colbuf;

/* A library-owned byte buffer. */
typedef struct {
    const uint8_t* data;
    uint64_t       size;
    int32_t        mem;
    int32_t        reserved_;
} cph_bytes;

/*
 * out[i] = col[row_ids[i] - id_base] for i < nrows (row_ids == NULL: a plain copy of
 * the column).  This is mergeRows (csvplus.go:571-583) column by column: a joined
 * table consists of the stream's columns (gathered through stream_row / probe_idx,
 * or used as they are when that is the identity) and of every index's columns
 * gathered through build_row; on a column-name collision the caller keeps the
 * stream's column (the stream value wins, :578-580).  row_ids are uint32
 * (id_bits 32) or uint64 (64) and live in the same memory space as the column.
 */
CPH_API int32_t cph_gather_rows(cph_ctx* ctx, const cph_strcol* col, const void* row_ids, int32_t id_bits,
                                uint64_t id_base, uint64_t nrows, int32_t out_mem, cph_colbuf** out);
CPH_API void    cph_colbuf_release(cph_colbuf* c);

/*
 * A column of the table an index was built over, in INDEX ORDER: out[p] = col[perm[p]] for the index's sorted
 * positions p (cph_gather_rows through cph_index_perm).  The reference's Index holds its rows sorted (createIndex,
 * csvplus.go:736) and Join reads index.impl.rows[first() + i]; a device pipeline that keeps the payload columns of a
 * build table in this order once (IndexOn time) consumes the sorted positions a Join reports (cph_join_chain_ex
 * CPH_CHAIN_POSITIONS, cph_stream_join_set_positions) directly as row subscripts — in cph_gather_rows, in
 * cph_csv_write_rows' cph_rowsel — and never needs the original row ids.  col must have the rows of the indexed table
 * (col->nrows > every perm value) and lives in host or device memory like the columns of cph_gather_rows.
 */
CPH_API int32_t cph_index_permute(cph_ctx* ctx, cph_index* index, const cph_strcol* col, int32_t out_mem, cph_colbuf** out);

/*
 * ToCsv (csvplus.go:379-406): the canonical serialisation — a header line (when
 * `header` != NULL: ncols names) and one record per row with the columns in the
 * given order, formatted as Go's encoding/csv Writer does with default settings
 * (',' separator, '\n' record end; a field is quoted iff it is `\.`, contains
 * ',', '"', '\r' or '\n', or starts with a Unicode space; '"' is doubled inside
 * quotes).  All columns must have the same number of rows.
 */
CPH_API int32_t cph_csv_write(cph_ctx* ctx, const cph_strcol* cols, int32_t ncols, const cph_strval* header,
                              int32_t out_mem, cph_bytes** out);

/*
 * Join(...).ToCsv(...) without materialising the joined table: output row i takes field c from row
 * sel[c].ids[i] - sel[c].base of cols[c] (sel == NULL or sel[c].ids == NULL: row i itself, and then
 * cols[c].nrows must equal nrows unless nrows is 0).  This is mergeRows (csvplus.go:571-583) folded into the writer:
 * the caller lists, per output column, the table it comes from (for a name present on both sides the
 * stream's column, :578-580) and the row-id array of that table from cph_join_chain / cph_join_probe.
 * Row ids live in the same memory space as their column and must be < cols[c].nrows.
 */
typedef struct {
    const void* ids;      /* uint32 or uint64 row ids, nrows entries; NULL = identity */
    int32_t     bits;     /* 32 or 64 */
    int32_t     reserved_;
    uint64_t    base;     /* subtracted from every id (e.g. the probe_base of a chunk) */
} cph_rowsel;

CPH_API int32_t cph_csv_write_rows(cph_ctx* ctx, const cph_strcol* cols, const cph_rowsel* sel, int32_t ncols, uint64_t nrows,
                                   const cph_strval* header, int32_t out_mem, cph_bytes** out);
CPH_API void    cph_bytes_release(cph_bytes* b);

/* ---- CSV ingest: bytes -> SoA string columns (csvplus.go:1080-1146) ----------- */
/*
 * Replaces the parse loop of Reader.Iterate (csv.NewReader + one map per line,
 * csvplus.go:1104-1131): the caller resolves the header to FIELD INDICES
 * (makeHeader, csvplus.go:1149-1206, stays on the host) and receives one
 * string column per requested index.  Semantics are Go's encoding/csv Reader
 * as csvplus configures it (csvplus.go:1088-1093): Comma, Comment,
 * TrimLeadingSpace, FieldsPerRecord; "\r\n" -> "\n"; empty and comment lines
 * skipped; `""` = literal quote.  LazyQuotes = 1 is rejected (CPH_ERR_INVALID),
 * and so is a comment line that contains a '"' (Go skips it uninterpreted; the
 * parallel quote-parity scan cannot).  A record with fewer fields than a
 * requested index yields "" for that column — callers that want the
 * reference's "missing column" error (csvplus.go:1121-1124) set
 * fields_per_record > 0, as csvplus itself does with NumFields.
 */
enum {
    CPH_CSV_ERR_BARE_QUOTE  = 1,  /* csv.ErrBareQuote: '"' inside an unquoted field          */
    CPH_CSV_ERR_QUOTE       = 2,  /* csv.ErrQuote: stray or unterminated '"' in a quoted field */
    CPH_CSV_ERR_FIELD_COUNT = 3   /* csv.ErrFieldCount                                        */
};

typedef struct {
    uint8_t  comma;               /* Reader.delimiter (csvplus.go:1088); ASCII, not '"' '\r' '\n' */
    uint8_t  comment;             /* Reader.comment   (csvplus.go:1089); 0 = none                 */
    uint8_t  trim_leading_space;  /* csvplus.go:1092                                              */
    uint8_t  lazy_quotes;         /* csvplus.go:1091; must be 0                                   */
    int32_t  fields_per_record;   /* csv.Reader.FieldsPerRecord: >0 exact, 0 = as the first record, <0 = any */
    uint64_t skip_records;        /* leading records to drop from the output (the header line; they are
                                     still parsed, validated and counted in error_record)          */
} cph_csv_options;

typedef struct {
    uint64_t   nrecords;          /* records returned: those before the first error, minus skip_records */
    uint64_t   error_record;      /* if error_kind != 0: 0-based record index (skipped records included) */
    int32_t    error_kind;        /* 0 or CPH_CSV_ERR_*: the first error in record order; the reference
                                     delivers the rows before it and then fails the same way           */
    int32_t    ncols;
    cph_strcol cols[CPH_MAX_KEY_COLS];   /* cols[i] = field col_index[i] of every record, in out_mem; offsets are
                                            32-bit when the text is smaller than 4 GiB, else 64-bit */
} cph_csv_table;

/* `data`/`size` = the whole CSV text in `mem` (CPH_MEM_HOST or CPH_MEM_DEVICE).
 * col_index[ncols] = 0-based field indices wanted, 1 <= ncols <= CPH_MAX_KEY_COLS. */
CPH_API int32_t cph_csv_parse(cph_ctx* ctx, const uint8_t* data, uint64_t size, int32_t mem, const cph_csv_options* opt,
                              const int32_t* col_index, int32_t ncols, int32_t out_mem, cph_csv_table** out);
CPH_API void    cph_csv_table_release(cph_csv_table* t);

/* ---- Find / SubIndex bounds (csvplus.go:870-891) ----------------------------- */

/* [*lower, *upper) = sorted positions whose leading key columns equal
 * `values` (nvalues <= key columns; 0 values = the whole index). */
CPH_API int32_t cph_index_find(cph_ctx* ctx, const cph_index* index, const cph_strval* values, int32_t nvalues,
                               uint64_t* lower, uint64_t* upper);

/* Many Find / SubIndex calls at once (csvplus.go:625-641 in a loop, e.g. BenchmarkSearchSmallSingleIndex,
 * csvplus_test.go:1104-1116): key k is values[k * nvalues .. (k + 1) * nvalues); lower[k] / upper[k] as cph_index_find.
 * One upload, ONE kernel launch and one download for the whole batch instead of a launch and two waits per key. */
CPH_API int32_t cph_index_find_many(cph_ctx* ctx, const cph_index* index, const cph_strval* values, int32_t nvalues,
                                    uint64_t nkeys, uint64_t* lower, uint64_t* upper);

/* ---- ResolveDuplicates support + persistence (csvplus.go:643-705, :810-867) ---- */

/*
 * Every maximal run of >= 2 equal keys, ascending: group g = sorted positions
 * [lower[g], upper[g]).  This is what dedup (csvplus.go:810-867) discovers one
 * group at a time — the adjacent-equal scans :815-819 / :851-855 give
 * lower[g]+1, the sort.Search :828-830 gives upper[g].  The resolve callback
 * and the compaction rule (:823-860, including which rows survive) stay with
 * the caller; cph_index_select then builds the compacted index.
 * Arrays are host memory owned by the library until cph_groups_release.
 */
typedef struct {
    uint64_t        ngroups;
    const uint64_t* lower;
    const uint64_t* upper;
} cph_groups;

CPH_API int32_t cph_index_dup_groups(cph_ctx* ctx, const cph_index* index, cph_groups** out);
CPH_API void    cph_groups_release(cph_groups* g);

/* New index = the rows at the given sorted positions of `index` (host array,
 * strictly ascending, < nrows; CPH_ERR_INVALID otherwise): index.rows[:dest]
 * after dedup's in-place compaction (csvplus.go:845-863).  Row ids (perm) keep
 * referring to the original build table. */
CPH_API int32_t cph_index_select(cph_ctx* ctx, const cph_index* index, const uint64_t* positions, uint64_t n,
                                 cph_index** out);

/* Index.WriteTo / LoadIndex (csvplus.go:655-705) for the device index: a flat
 * little-endian file with the key codec, sorted codes and perm.  NOT a gob
 * stream and without row payload: the row table stays with the caller (the
 * reference's file holds the rows because there the rows ARE the index).  A
 * short write removes the file, as the reference does (:663-671). */
CPH_API int32_t cph_index_save(cph_ctx* ctx, const cph_index* index, const char* path);
CPH_API int32_t cph_index_load(cph_ctx* ctx, const char* path, cph_index** out);

/* ---- introspection (for benchmarks / roofline accounting) -------------------- */

typedef struct {
    uint64_t nrows;
    int32_t  nkeycols;
    int32_t  key_positions;   /* total byte positions encoded (sum of per-column max length) */
    int32_t  code_words;      /* 64-bit (or one 32-bit) words per encoded key                */
    int32_t  code_bits;       /* significant bits of the most significant..least words, summed */
    int32_t  key_bytes;       /* bytes per key the sort kernels move (4 or 8)                 */
    int32_t  sort_passes;     /* radix scatter passes executed                                 */
    int32_t  direct_table;    /* 1 when the probe uses the direct-address table                */
    int32_t  dict_entries;    /* entries of the group dictionaries (0: per-position alphabets only)  */
    uint64_t table_entries;   /* entries of the direct-address table (0 if none planned)       */
    int32_t  lookup_built;    /* lookup structures BUILT so far (bits): 8 = rank table (positions), 1 = 8-byte table, 2 = 4-byte row table,
                                 4 = hash table.  direct_table / table_entries only say one is PLANNED: the
                                 structures are built by the first Join that uses them (or cph_index_prepare_join) */
    int32_t  hash_mode;       /* 0 none; 1 one code word per entry, 2 up to three words, 3 64-bit tag + verification */
    uint64_t hash_bytes;      /* size of the hash table                                                         */
    int32_t  build_path;      /* 0: general path (statistics, host codec, encode, multi-launch radix sort);
                                 1: the one-launch build of small tables (ctx option "small_build_rows", default 8192);
                                 2: host-formed codes — the key column came in host memory and only its 4-byte codes were
                                    uploaded (ctx option "host_build") */
    int32_t  split;           /* 0: none; else 0x100 * (1 + key column that is cut) + the delimiter byte: the key codec codes that
                                 column as (prefix through its first delimiter, suffix) — whole-value dictionary + per-position
                                 suffix (round 4; dict_entries then counts the prefixes) */
} cph_index_info;

CPH_API int32_t cph_index_get_info(const cph_index* index, cph_index_info* info);

/*
 * Join looks a probe row's key up in a structure that is not part of the Index (csvplus.go:612-614: the sorted rows
 * ARE the index) and is therefore built by the first Join that wants it: a direct-address table over the key codes
 * when the code space is dense (<= 24 codes per row) — for a duplicate-free index asked for sorted positions or for
 * bounds only (cph_join_chain_ex CPH_CHAIN_POSITIONS, cph_join_probe with want_pairs = 0) a RANK table instead: presence
 * bits + a running count per 32 codes, 1/16 the size of the 4-byte row table —, a hash table over the codes otherwise
 * (one 64-byte sector per probe row for any key: random ids, hashes, several columns, keys of any length); a PREFIX join (fewer columns than
 * the index has, csvplus.go:546-550) needs the order and searches the sorted codes, as does every Join when the
 * structure cannot be allocated.  This call builds the structure NOW — `chained` = 1: the one cph_join_chain /
 * cph_stream_join use (4-byte row table for a duplicate-free index), 2: the one cph_join_chain_ex uses with
 * CPH_CHAIN_POSITIONS (the rank table), 0: the one cph_join_probe uses for pairs — so that
 * the first Join does not pay for it.
 *
 * Sharing an index between ctxs: the structures are built on the INDEX's ctx (its stream, its pool) whatever ctx
 * runs the Join, and a Join on another ctx of the same device orders its stream behind that build.  Two threads
 * with a ctx each may join against one index at the same time: the lazy build is serialised per index and the
 * index ctx's device pool is locked (since round 4; cph_index_prepare_join remains the way to keep the build out
 * of the first Join).  The index's OWN ctx must not be destroyed — nor the index itself — while another thread joins
 * against it.
 */
CPH_API int32_t cph_index_prepare_join(cph_index* index, int32_t chained);

/*
 * Per-kernel timing for roofline accounting.  While enabled, every kernel launch
 * of this ctx is bracketed by two HIP events on the ctx's stream.
 * cph_ctx_profile_read synchronises the stream and returns, per kernel name, the
 * number of launches, their summed duration and their summed ALGORITHMIC bytes
 * (the per-launch byte model documented in DESIGN.md; 0 where the library cannot
 * know the input's value bytes).  *n receives the number of distinct kernels
 * (may exceed cap).  reset != 0 clears the statistics afterwards.
 */
typedef struct {
    char     name[48];
    uint64_t launches;
    double   total_ms;
    double   algo_bytes;
} cph_kernel_stat;

CPH_API int32_t cph_ctx_profile(cph_ctx* ctx, int32_t enable);
/* The two events around a launch cost ~10 us of stream time each: a loop of many short kernels measurably slows
 * down under cph_ctx_profile.  This variant times ONLY the launches of `kernel_name` (NULL: profiling off), so a
 * benchmark can time its dominant kernel inside the timed region without disturbing the region. */
CPH_API int32_t cph_ctx_profile_only(cph_ctx* ctx, const char* kernel_name);
CPH_API int32_t cph_ctx_profile_read(cph_ctx* ctx, cph_kernel_stat* out, int32_t cap, int32_t* n, int32_t reset);

/*
 * What this box's memory system sustains for the two access patterns the Join kernels are made of (a measurement
 * utility, not part of the hot path): kind 0 = streaming copy, `bytes` read + `bytes` written with 16 bytes per lane;
 * kind 1 = random gather of 4-byte entries, `n` lookups into a table of `bytes` bytes, 4 lookups in flight per thread,
 * the indices (4 B) streamed in and the values (4 B) streamed out — one direct-table step of the chained-join kernel
 * without any key decoding.  *ms = average of `reps` launches after two warm-up launches (HIP events on the ctx stream).
 */
CPH_API int32_t cph_calibrate(cph_ctx* ctx, int32_t kind, uint64_t bytes, uint64_t n, int32_t reps, double* ms);

/* Library version, e.g. "csvplus_hip 0.1 (gfx950)". */
CPH_API const char* cph_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CSVPLUS_HIP_H */
