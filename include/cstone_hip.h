/* cstone_hip.h -- C ABI of the MI355X-native cornerstone-octree hot path (libcstone_hip.so)
 *
 * This is the drop-in boundary.  The reference (cornerstone-octree) has no FFI: its GPU flavour is
 * reached through a LINK SEAM of `template<...> extern void fooGpu(...)` declarations whose
 * definitions live in the .cu files of its cstone_gpu library.  Every entry point below replaces
 * one of those seam functions (cited as R/<file>:<line>, R = /root/reference/include/cstone) with
 * a plain-C signature: device pointers + sizes, no C++/torch types.  The C++20 header layer in
 * cornerstone-octree_amd/include/cstone_amd/ turns these back into the reference's templates
 * (same names and argument meaning), see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = error (CSTONE_E_*), message via cstone_hip_last_error
 *   - all array pointers are DEVICE pointers unless the parameter name ends in _host
 *   - work is enqueued on the context's HIP stream and is asynchronous unless a host result is
 *     returned (those calls synchronise the stream)
 *   - key_bits in {32,64} selects KeyType = uint32_t | uint64_t (R/tree/definitions.h:46-83)
 *   - real_bits in {32,64} selects float | double coordinates
 *   - curve: CSTONE_MORTON | CSTONE_HILBERT (the reference fixes this at build time through the
 *     SfcKind alias, R/sfc/sfc.hpp:53-55; here it is a run-time argument)
 *   - TreeNodeIndex = int32, LocalIndex = uint32 (R/tree/definitions.h:41-43)
 *   - one context per (host thread, device); calls on one context must be serialised by the caller
 */
#ifndef CSTONE_HIP_H
#define CSTONE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

#define CSTONE_MORTON 0
#define CSTONE_HILBERT 1

#define CSTONE_OK 0
#define CSTONE_E_ARG (-1)      /* invalid argument (bad key_bits, null pointer, size too large ...) */
#define CSTONE_E_CAPACITY (-2) /* caller-provided buffer too small; required size reported through out-params */
#define CSTONE_E_HIP (-3)      /* a HIP runtime call failed */
#define CSTONE_E_INTERNAL (-4) /* device-side consistency check failed (bounded spin expired, stack overflow ...) */

    /* Global coordinate bounding box, POD image of cstone::Box<T> (R/sfc/box.hpp:112-191).
     * lim = {xmin,xmax,ymin,ymax,zmin,zmax}; the library derives lengths and inverse lengths in the
     * precision selected by real_bits exactly as the Box<T> constructor does (1/(max-min), :135).
     * bc[d]: 0 open, 1 periodic, 2 fixed (BoundaryType, R/sfc/box.hpp:97-102). */
    typedef struct cstone_box
    {
        double lim[6];
        int32_t bc[3];
        int32_t pad_;
    } cstone_box;

    typedef struct cstone_hip_ctx cstone_hip_ctx;

    /* ---------------------------------------------------------------------------------------------
     * runtime: replaces R/cuda/device_vector.h, cuda_stubs.h:48-57 (memcpyH2D/D2H/D2D, syncGpu),
     * errorcheck.cuh:30-42 (we return codes instead of exit()).
     * ------------------------------------------------------------------------------------------- */
    /* private_stream != 0: the context creates (and owns) a non-blocking stream of its own;
     * private_stream == 0: work goes to `stream`, an existing hipStream_t such as torch's current
     *                      stream (NULL = the device's default stream) */
    int cstone_hip_ctx_create(cstone_hip_ctx** out, int device, void* stream, int private_stream);
    int cstone_hip_ctx_destroy(cstone_hip_ctx* ctx);
    int cstone_hip_ctx_sync(cstone_hip_ctx* ctx);
    /* text of the last error of this context; ctx == NULL: the last failed cstone_hip_ctx_create of this thread */
    const char* cstone_hip_last_error(cstone_hip_ctx* ctx);
    /* number of compute units / wavefront size of the context's device (R/cuda/gpu_config.cuh:41-59) */
    int cstone_hip_device_info(cstone_hip_ctx* ctx, int* num_cu, int* wave_size);

    int cstone_hip_malloc(cstone_hip_ctx* ctx, void** ptr, size_t bytes);
    int cstone_hip_free(cstone_hip_ctx* ctx, void* ptr);
    int cstone_hip_memcpy_h2d(cstone_hip_ctx* ctx, void* dst, const void* src_host, size_t bytes);
    int cstone_hip_memcpy_d2h(cstone_hip_ctx* ctx, void* dst_host, const void* src, size_t bytes);
    int cstone_hip_memcpy_d2d(cstone_hip_ctx* ctx, void* dst, const void* src, size_t bytes);
    int cstone_hip_memset(cstone_hip_ctx* ctx, void* dst, int value, size_t bytes);

    /* Stage timers (HIP events on the context's stream).  When enabled, the library brackets each
     * launch of the named stages with events; totals are read back with cstone_hip_profile_get.
     * Stage ids: CSTONE_STAGE_*.  Adds two event records per bracket; off by default. */
#define CSTONE_STAGE_ENCODE 0
#define CSTONE_STAGE_SORT_HIST 1
#define CSTONE_STAGE_SORT_PASS 2 /* one onesweep digit pass = one launch */
#define CSTONE_STAGE_GATHER 3
#define CSTONE_STAGE_NODE_COUNTS 4
#define CSTONE_STAGE_REBALANCE 5
#define CSTONE_STAGE_LINK_OCTREE 6
#define CSTONE_STAGE_HALOS 7
#define CSTONE_STAGE_NEIGHBORS 8
#define CSTONE_STAGE_MINMAX 9
#define CSTONE_STAGE_SORT_PASS_IOTA 10 /* a digit pass that produces the positions instead of reading values: K + (K+4) B/pair */
#define CSTONE_STAGE_RESORT_BINS 11   /* incremental re-sort of Domain::sync: leaf table, mover bins (csrc/resort.hpp) */
#define CSTONE_STAGE_RESORT_LEAVES 12 /* ... its pass over the leaves: K read, K + 4 written per particle */
#define CSTONE_STAGE_GATHER_H 13      /* gather of h fused with the halo radii of Domain::sync: 4 + 2 T bytes per particle */
#define CSTONE_STAGE_PLACE 14         /* multi-rank sync: keys, x, y, z, h of the kept particles to their final slots in one
                                        pass (placeColumnsKernel): 4 + 2 K + 8 T bytes per particle (+ 4 with a merge) */
#define CSTONE_NUM_STAGES 16
    /* on: 0 off, 1 every stage, 2 only ENCODE, SORT_PASS(_IOTA), RESORT_LEAVES, GATHER(_H), HALOS, NEIGHBORS (the kernels that
     * move the particle arrays: eight brackets per sync instead of forty) */
    int cstone_hip_profile_enable(cstone_hip_ctx* ctx, int on);
    /* on != 0: every stage of every call is also wrapped in a roctx range "cstone:<stage>" (roctxRangePushA / Pop of
     * librocprofiler-sdk-roctx, opened on first use): `rocprofv3 --marker-trace --kernel-trace` then shows the stage table
     * of a sync without the event brackets above (the reference's tracing hooks).  Independent of
     * cstone_hip_profile_enable. */
    int cstone_hip_profile_markers(cstone_hip_ctx* ctx, int on);
    int cstone_hip_profile_reset(cstone_hip_ctx* ctx);
    /* synchronises the stream; total_ms and launches accumulated since the last reset */
    int cstone_hip_profile_get(cstone_hip_ctx* ctx, int stage, double* total_ms, int* launches);
    /* min / median / max of the individual brackets of a stage since the last reset (the first 8192 are kept) */
    int cstone_hip_profile_get_spread(cstone_hip_ctx* ctx, int stage, double* min_ms, double* median_ms, double* max_ms);

    /* ---------------------------------------------------------------------------------------------
     * SFC keys: replaces computeSfcKeysGpu (R/sfc/sfc_gpu.h:37-38, kernel R/sfc/sfc_gpu.cu:39-57).
     * keys[i] = sfc3D(x[i],y[i],z[i],box) unless keys[i] already holds the remove marker
     * 2^(3*maxLevel) (R/sfc/sfc.hpp:284-291).  Bit-exact with the reference's CPU path.
     * ------------------------------------------------------------------------------------------- */
    int cstone_hip_compute_sfc_keys(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* x,
                                    const void* y, const void* z, void* keys, size_t n, const cstone_box* box_host);

    /* computeSfcKeys + setMapFromCodes as GlobalAssignment::assign issues them (R/domain/assignment.hpp:81-86) in one
     * call: keys[i] as compute_sfc_keys, then keys sorted in place and ordering[n] = the sorting permutation.  The digits
     * of the sort are counted while the keys are still in the encode kernel's registers. */
    int cstone_hip_sfc_keys_and_ordering(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* x,
                                         const void* y, const void* z, void* keys, uint32_t* ordering, size_t n,
                                         const cstone_box* box_host, void* keys_alt, uint32_t* values_alt, void* temp,
                                         size_t temp_bytes);

    /* ---------------------------------------------------------------------------------------------
     * sort: replaces sortByKeyGpu / sortByKeyTempStorage (R/primitives/primitives_gpu.h:93-100,
     * R/primitives/primitives_gpu.cu:328-369,383-386) and sequenceGpu (:286).
     * Stable ascending LSD radix sort of (key, uint32 value) pairs over ALL key bits; the sorted
     * sequence ends up in keys/values (like the reference, which copies back from the alt buffer).
     * n < 2^30.  keys_alt[n], values_alt[n] and temp (>= cstone_hip_sort_pairs_temp_bytes) are
     * caller-provided scratch; pass NULL for all three to let the context's arena provide them.
     * ------------------------------------------------------------------------------------------- */
    size_t cstone_hip_sort_pairs_temp_bytes(int key_bits, size_t n);
    /* GpuSfcSorter::setMapFromCodes (R/primitives/gather.cuh:73-85) = sequenceGpu + sortByKeyGpu in one: sorts keys and
     * writes the sorting permutation to ordering[n] (its previous content is ignored: the first digit pass produces
     * the positions instead of reading them).  All scratch arrays are required. */
    int cstone_hip_sort_keys_ordering(cstone_hip_ctx* ctx, int key_bits, void* keys, uint32_t* ordering, size_t n,
                                      void* keys_alt, uint32_t* values_alt, void* temp, size_t temp_bytes);
    int cstone_hip_sort_pairs(cstone_hip_ctx* ctx, int key_bits, void* keys, uint32_t* values, size_t n, void* keys_alt,
                              uint32_t* values_alt, void* temp, size_t temp_bytes);
    int cstone_hip_sequence_u32(cstone_hip_ctx* ctx, uint32_t* out, size_t n, uint32_t init);

    /* gatherGpu / scatterGpu (R/primitives/primitives_gpu.h:48-56): dst[i] = src[map[i]] resp.
     * dst[map[i]] = src[i] for elements of elem_bytes in {1,2,4,8,12,16,24,32} */
    int cstone_hip_gather(cstone_hip_ctx* ctx, int elem_bytes, const uint32_t* map, size_t n, const void* src,
                          void* dst);
    int cstone_hip_scatter(cstone_hip_ctx* ctx, int elem_bytes, const uint32_t* map, size_t n, const void* src,
                           void* dst);
    /* gather of 1..4 arrays of equal element size (4, 8 or 16 bytes) through the SAME map, which is read once:
     * dst[a][i] = src[a][map[i]] (what gatherArrays, R/domain/layout.hpp:203-239, does array after array); no
     * destination may be one of the sources */
    int cstone_hip_gather_multi(cstone_hip_ctx* ctx, int elem_bytes, const uint32_t* map, size_t n,
                                const void* const* src, void* const* dst, int num_arrays);

    /* gatherScatter (R/primitives/gather.hpp:120-131): dst[map_out[i]] = src[map_in[i]], element sizes as gather */
    int cstone_hip_gather_scatter(cstone_hip_ctx* ctx, int elem_bytes, const uint32_t* map_in, const uint32_t* map_out,
                                  size_t n, const void* src, void* dst);
    /* positions of the elements of two sorted key runs in their stable merge (ties: run a first), plus offset:
     * the re-sort of GlobalAssignment::distribute (R/domain/assignment.hpp:139-158) for a kept range that is already
     * sorted and newcomers that have been sorted among themselves */
    int cstone_hip_merge_positions(cstone_hip_ctx* ctx, int key_bits, const void* a, size_t na, const void* b,
                                   size_t nb, uint32_t offset, uint32_t* pos_a, uint32_t* pos_b);

    /* MinMaxGpu (R/primitives/primitives_gpu.h:58-62): out_host = {min, max} as doubles (exact for float) */
    int cstone_hip_minmax(cstone_hip_ctx* ctx, int real_bits, const void* x, size_t n, double* out2_host);
    /* the same for 1..3 equally long arrays (the x, y, z of the bounding box, R/sfc/box_mpi.hpp:40-70) in one launch and
     * one read-back: out_host = {min0, max0, min1, max1, ...} */
    int cstone_hip_minmax_arrays(cstone_hip_ctx* ctx, int real_bits, const void* const* arrays, int num_arrays,
                                 size_t n, double* out_host);

    /* exclusiveScanGpu / inclusiveScanGpu on uint32/int32 (R/primitives/primitives_gpu.h:103-115).
     * exclusive: out[i] = init + sum(in[0..i)), n outputs.  in == out allowed. */
    int cstone_hip_exclusive_scan_u32(cstone_hip_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, uint32_t init);
    int cstone_hip_inclusive_scan_u32(cstone_hip_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n);
    /* the exclusive scan with its grand total behind it: out[i] = sum(in[0..i)) for i <= n (n + 1 outputs; in != out) --
     * offsets from counts in one call (the reference's fill(0) + inclusive scan to out + 1, e.g. computeNodeLayout,
     * R/domain/layout.hpp:150-165) */
    int cstone_hip_offsets_from_counts_u32(cstone_hip_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n);

    /* lowerBoundGpu, range form (R/primitives/primitives_gpu.h:70-71): result[q] = index of the first element of
     * the sorted keys[n] that is >= values[q] (unsigned compare); values and result are device arrays.
     * Used for createSendRanges (R/domain/domaindecomp.hpp:218-230). */
    int cstone_hip_lower_bound(cstone_hip_ctx* ctx, int key_bits, const void* keys, size_t n, const void* values,
                               int num_values, uint64_t* result);

    /* ---------------------------------------------------------------------------------------------
     * the small primitives of the seam (R/primitives/primitives_gpu.h:36-124, Thrust one-liners in the reference)
     * fill          : fillGpu (:39-40), dst[i] = *value_host for elements of 1, 4 or 8 bytes
     * scale         : scaleGpu (:42-43), data[i] *= factor (float | double)
     * increment     : incrementGpu (:45-46), out[i] = in[i] + value for uint32 | uint64 (elem_bits)
     * count_equal   : countGpu (:123-124), number of elements equal to value (elements of 32 | 64 bits)
     * reduce_sum    : reduceGpu (:82-83), init + sum of uint32 | uint64 elements in 64-bit arithmetic
     * max_norm_square : maxNormSquareGpu (:64-65), max of x^2 + y^2 + z^2 evaluated in real_bits precision
     * segment_max   : segmentMax (:79-80), out[s] = max(0, in[segments[s] .. segments[s+1])): every segment starts at 0
     *                 like the reference's kernel (:241-259), an empty segment gives 0; in/out float|double, segments
     *                 uint32|uint64
     * gather_ranges : gatherRanges (R/halos/gather_halos_gpu.h): buffer[i] = src[range_offsets[r] + i - range_scan[r]]
     *                 for the range r with range_scan[r] <= i < range_scan[r+1]; indices uint32 | uint64 (index_bits),
     *                 elements of 1..32 bytes as gather
     * lower_bound_value : lowerBoundGpu, scalar form (:67-68); kind 0 u32, 1 u64, 2 i32, 3 i64, 4 f32
     * sort_keys     : sortGpu (:91-92), ascending keys-only sort (the key buffer of the reference's signature is not
     *                 needed: scratch comes from the context)
     * ------------------------------------------------------------------------------------------- */
    int cstone_hip_fill(cstone_hip_ctx* ctx, int elem_bytes, void* dst, size_t n, const void* value_host);
    int cstone_hip_scale(cstone_hip_ctx* ctx, int real_bits, void* data, size_t n, double factor);
    int cstone_hip_increment(cstone_hip_ctx* ctx, int elem_bits, const void* in, void* out, size_t n, uint64_t value);
    /* out[i] = in[i + 1] - in[i], i < n (in: n + 1 offsets; in != out): the group sizes computeGroupSplits hands back in
     * numSplitsPerGroup (R/traversal/groups_gpu.cu:108-117) */
    int cstone_hip_adjacent_difference_u32(cstone_hip_ctx* ctx, const uint32_t* in, size_t n, uint32_t* out);
    /* up to three small u32 tables into ONE output array (then one copy brings them to the host): out[i] = a[map[i]]
     * for i < n_a (zeros when a is null), out[n_a + j] = b[map[n_a + j]] for j < n_b, out[n_a + n_b + k] = c[k] for
     * k < n_c.  The read-back at the end of Halos::computeLayout (R/domain/layout.hpp:175-190: where the halos of each
     * peer arrive, how many go out to each, the level ranges of the tree) */
    int cstone_hip_gather_tables_u32(cstone_hip_ctx* ctx, const uint32_t* map, const uint32_t* a, size_t n_a,
                                     const uint32_t* b, size_t n_b, const uint32_t* c, size_t n_c, uint32_t* out);
    int cstone_hip_count_equal(cstone_hip_ctx* ctx, int elem_bits, const void* data, size_t n, uint64_t value,
                               uint64_t* count_host);
    int cstone_hip_reduce_sum(cstone_hip_ctx* ctx, int elem_bits, const void* data, size_t n, uint64_t init,
                              uint64_t* sum_host);
    int cstone_hip_max_norm_square(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y, const void* z,
                                   size_t n, double* out_host);
    int cstone_hip_segment_max(cstone_hip_ctx* ctx, int in_bits, int out_bits, int index_bits, const void* in,
                               const void* segments, size_t num_segments, void* out);
    int cstone_hip_gather_ranges(cstone_hip_ctx* ctx, int elem_bytes, int index_bits, const void* range_scan,
                                 const void* range_offsets, int num_ranges, const void* src, void* buffer,
                                 size_t buffer_size);
    /* gatherRanges (R/halos/gather_halos_gpu.cu:26-40) of up to four equally laid out arrays of 4- or 8-byte elements
     * into ROWS, rows[i * num_arrays + a] = src[a][range_offsets[r] + i - range_scan[r]], and the receiving side,
     * dst[a][dst_offset + i] = rows[i * num_arrays + a]: one message per peer for all arrays of a halo exchange
     * (the reference sends one message per array and peer, R/halos/exchange_halos_gpu.cuh:71-115) */
    int cstone_hip_gather_ranges_rows(cstone_hip_ctx* ctx, int elem_bytes, int num_arrays, const uint32_t* range_scan,
                                      const uint32_t* range_offsets, int num_ranges, const void* const* src, void* rows,
                                      size_t num_rows);
    int cstone_hip_scatter_rows(cstone_hip_ctx* ctx, int elem_bytes, int num_arrays, const void* rows, size_t num_rows,
                                void* const* dst, size_t dst_offset);
    int cstone_hip_lower_bound_value(cstone_hip_ctx* ctx, int kind, const void* data, size_t n, const void* value_host,
                                     uint64_t* index_host);
    int cstone_hip_sort_keys(cstone_hip_ctx* ctx, int key_bits, void* keys, size_t n);
    /* the remaining instantiations of the reference's list (R/primitives/primitives_gpu.cu:214-238,395-437,88-103) on
     * the device: lowerBoundGpu with 32-bit results, sequenceGpu<uint64_t>, exclusive/inclusiveScanGpu with 64-bit sums
     * of 32-bit values (not in place) */
    int cstone_hip_lower_bound_u32(cstone_hip_ctx* ctx, int key_bits, const void* keys, size_t n, const void* values,
                                   int num_values, uint32_t* result);
    int cstone_hip_sequence_u64(cstone_hip_ctx* ctx, uint64_t* out, size_t n, uint64_t init);
    int cstone_hip_scan_u32_to_u64(cstone_hip_ctx* ctx, const uint32_t* in, uint64_t* out, size_t n, uint64_t init,
                                   int inclusive);

    /* ---------------------------------------------------------------------------------------------
     * cornerstone leaf array (R/tree/csarray_gpu.h:56-88, R/tree/update_gpu.cuh:59-82)
     * ------------------------------------------------------------------------------------------- */
    /* computeNodeCountsGpu: counts[i] = min(#keys in [tree[i],tree[i+1]), max_count); keys sorted */
    int cstone_hip_compute_node_counts(cstone_hip_ctx* ctx, int key_bits, const void* tree, uint32_t* counts,
                                       int num_nodes, const void* keys, size_t n, uint32_t max_count);
    /* the same with a starting point per leaf boundary (useCountsAsGuess, R/tree/csarray.hpp:117-186):
     * guess_positions[num_nodes + 1] = where tree[i] was found last time (e.g. the previous layout); any values are
     * allowed, good ones end the search after two or three probes instead of log2(n); NULL = plain binary search */
    int cstone_hip_compute_node_counts_guided(cstone_hip_ctx* ctx, int key_bits, const void* tree, uint32_t* counts,
                                              int num_nodes, const void* keys, size_t n, uint32_t max_count,
                                              const uint32_t* guess_positions);
    /* computeNodeOpsGpu: node_ops[num_nodes+1] <- exclusive scan of the rebalance decisions
     * (R/tree/csarray.hpp:288-310); *new_num_nodes_host = node_ops[num_nodes];
     * *converged_host = 1 iff every decision was "keep" */
    int cstone_hip_compute_node_ops(cstone_hip_ctx* ctx, int key_bits, const void* tree, int num_nodes,
                                    const uint32_t* counts, uint32_t bucket_size, int32_t* node_ops,
                                    int* new_num_nodes_host, int* converged_host);
    /* rebalanceTreeGpu: new_tree[new_num_nodes+1] from tree and scanned node_ops */
    int cstone_hip_rebalance_tree(cstone_hip_ctx* ctx, int key_bits, const void* tree, int num_nodes,
                                  int new_num_nodes, const int32_t* node_ops, void* new_tree);
    /* updateOctreeGpu: one rebalance step + recount on device-resident buffers with capacity
     * cap_leaves (tree: cap_leaves+1 keys).  *num_leaves_host is updated.  CSTONE_E_CAPACITY if the
     * new tree does not fit (then *num_leaves_host holds the required leaf count, buffers unchanged). */
    int cstone_hip_update_octree(cstone_hip_ctx* ctx, int key_bits, const void* keys, size_t n, uint32_t bucket_size,
                                 void* tree, uint32_t* counts, int* num_leaves_host, int cap_leaves,
                                 uint32_t max_count, int* converged_host);
    /* computeOctree (R/tree/csarray.hpp:453-466): start at the root, update until converged */
    int cstone_hip_compute_octree(cstone_hip_ctx* ctx, int key_bits, const void* keys, size_t n, uint32_t bucket_size,
                                  void* tree, uint32_t* counts, int* num_leaves_host, int cap_leaves,
                                  uint32_t max_count, int* iterations_host);

    /* createBinaryTreeGpu (R/tree/btree.cuh:41-52): the binary radix tree (Karras 2012) over the
     * num_nodes + 1 keys of a cornerstone leaf array, num_nodes internal nodes of the reference's BinaryNode<KeyType>
     * layout {int32 child[2]; KeyType prefix} (12 bytes for 32-bit keys, 16 for 64-bit keys).  A child that is a leaf is
     * stored as (key index - 2^31) (R/tree/btree.hpp:48-66); prefix = common key prefix with the placeholder bit. */
    int cstone_hip_create_binary_tree(cstone_hip_ctx* ctx, int key_bits, const void* tree, int num_nodes,
                                      void* binary_nodes);

    /* ---------------------------------------------------------------------------------------------
     * linked octree: replaces buildOctreeGpu (R/tree/octree_gpu.h:47, R/tree/octree_gpu.cu:152-174).
     * Sizes with L = num_leaves, I = (L-1)/7, M = L+I:
     *   prefixes[M] child_offsets[M+1] parents[max(1,(M-1)/8)] level_range[maxLevel+2]
     *   internal_to_leaf[M] leaf_to_internal[M]          (OctreeView, R/tree/octree.hpp:280-293)
     * upsweepSumGpu (R/tree/octree_gpu.h:50): saturating u32 sum of 8 children, bottom-up.
     * ------------------------------------------------------------------------------------------- */
    int cstone_hip_build_octree(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves, void* prefixes,
                                int32_t* child_offsets, int32_t* parents, int32_t* level_range,
                                int32_t* internal_to_leaf, int32_t* leaf_to_internal);
    int cstone_hip_upsweep_sum(cstone_hip_ctx* ctx, int num_levels_plus2, const int32_t* level_range,
                               const int32_t* child_offsets, uint32_t* counts);
    /* the same two with a bound the caller has on the level of the deepest leaf (e.g. the deepest level of the tree
     * this one was rebalanced from, plus one): the node keys are sorted over 3 * deepest_level + 1 bits only and the
     * levels below the bound are not launched.  A bound that is too small is an error of the caller (undetected). */
    int cstone_hip_build_octree_bounded(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves,
                                        void* prefixes, int32_t* child_offsets, int32_t* parents, int32_t* level_range,
                                        int32_t* internal_to_leaf, int32_t* leaf_to_internal, int deepest_level);
    int cstone_hip_upsweep_sum_bounded(cstone_hip_ctx* ctx, int num_levels_plus2, const int32_t* level_range,
                                       const int32_t* child_offsets, uint32_t* counts, int deepest_level);
    /* computeGeoCentersGpu (R/focus/source_center_gpu.h): centers/sizes [M][3] reals */
    int cstone_hip_node_centers(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                                int num_nodes, const cstone_box* box_host, void* centers, void* sizes);

    /* ---------------------------------------------------------------------------------------------
     * halo discovery: replaces segmentMax + scaleGpu (R/primitives/primitives_gpu.h:79-80,38-39) as
     * used by Halos::discover (R/halos/halos.hpp:128-189) and findHalosGpu
     * (R/traversal/collisions_gpu.h:57-66, kernel R/traversal/collisions_gpu.cu:40-104).
     * halo_radii: radii[i] = float(max(h[layout[i-first]..layout[i-first+1])) * 2 * ext) for
     *             i in [first,last) (0 if the leaf is empty), 0 elsewhere; layout = last-first+1 offsets
     *             (this is the CPU branch's rounding, halos.hpp:176 -- parity is defined vs the CPU path)
     * find_halos: flags[num_leaves] must be pre-zeroed by the caller (fillGpu in the reference);
     *             flags[l] = 1 for every leaf l outside [leaves[first],leaves[last]) whose box
     *             overlaps the radius-dilated box of a leaf in [first,last)
     * ------------------------------------------------------------------------------------------- */
    int cstone_hip_halo_radii(cstone_hip_ctx* ctx, int h_bits, const void* h, const uint32_t* layout, int first,
                              int last, int num_leaves, float ext, float* radii);
    int cstone_hip_find_halos(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                              const int32_t* child_offsets, const int32_t* internal_to_leaf, const void* leaves,
                              const float* radii, const cstone_box* box_host, int first, int last, int32_t* flags);

    /* Building blocks of the multi-rank halo exchange with OWNER-SIDE discovery: instead of
     * keeping a locally-essential copy of remote tree structure (the R/focus headers), a rank exports the dilated boxes of its
     * boundary leaves and every owner answers on its own, finest tree.
     * halo_boxes   : for leaves [first,last): boxes[(i-first)*8 + 0..5] = {xlo,xhi,ylo,yhi,zlo,zhi} of
     *                makeHaloBox (R/traversal/boxoverlap.hpp:159-182), [6] = 1 if the box is NOT contained in the own
     *                key range [leaves[first], leaves[last]) (containedIn, :95-115), [7] = 0
     * find_overlaps: flags[l] |= 1 << (record[7] & 31) for every leaf l in [first,last) whose box overlaps
     *                (periodic-aware, :42-82) one of the num_boxes 8-int records with record[6] != 0; flags must be
     *                pre-zeroed.  record[7] = 0 (what halo_boxes writes) gives plain 0/1 flags; a caller that serves
     *                the boxes of up to 32 ranks in one traversal stores the exporting rank there */
    int cstone_hip_halo_boxes(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* leaves,
                              const float* radii, const cstone_box* box_host, int first, int last, int32_t* boxes);
    /* halo_boxes with proof: record[6] = 1 only for boxes that overlap a leaf of this linked tree OUTSIDE the key range
     * [leaves[first], leaves[last]) -- the walk of findHalos (R/traversal/collisions.hpp:79-105) stopped at the first
     * such leaf.  The enclosing-node test of halo_boxes (containedIn, :91-98) also lets through every box that
     * straddles a coarse octree boundary deep inside the own range; an exporter wants none of those. */
    int cstone_hip_halo_boxes_foreign(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                                      const int32_t* child_offsets, const int32_t* internal_to_leaf,
                                      const void* leaves, const float* radii, const cstone_box* box_host, int first,
                                      int last, int32_t* boxes);
    int cstone_hip_find_overlaps(cstone_hip_ctx* ctx, int curve, int key_bits, const void* prefixes,
                                 const int32_t* child_offsets, const int32_t* internal_to_leaf, const void* leaves,
                                 const int32_t* boxes, int num_boxes, int first, int last, int32_t* flags);

    /* ---------------------------------------------------------------------------------------------
     * focus tree (locally essential tree): the GPU seam of R/focus/rebalance_gpu.h:40-81, markMacsGpu
     * (R/traversal/collisions_gpu.h:68-77), countSfcGapsGpu / fillSfcGapsGpu (R/tree/csarray_gpu.h:78-88) and the node
     * spheres of R/focus/source_center_gpu.h:50-89.  Arrays indexed by node follow the linked octree (prefixes,
     * child_offsets, parents of cstone_hip_build_octree); macs / markings are char[num_nodes].
     * rebalance_decision_essential : node_ops[i] in {0 merge, 1 keep, 8 split} from counts and MAC flags
     *                 (mergeCountAndMacOp, R/focus/rebalance.hpp:50-79), focus = keys [focus_start, focus_end)
     * mac_refine_decision : node_ops[leaf] = 8 for leaves outside [focus_first, focus_last) whose MAC flag is set
     *                 (and that can still be split), 1 otherwise (macRefineOp, :81-88)
     * protect_ancestors : a 0 (merge) becomes the op of the closest ancestor with a non-zero op if the node is that
     *                 ancestor's left-most descendant (nzAncestorOp, :113-131); *converged_host = 1 iff every op is 1
     * enforce_keys  : makes sure the tree keeps / gets a node boundary at each of the num_forced_keys keys (device
     *                 array): cancels merges of the supporting ancestors and requests a split by ONE level
     *                 (enforceKeySingle, :199-250); *status_host = max over the keys of ResolutionStatus (:186-196:
     *                 0 converged, 1 cancelMerge, 2 rebalance, 3 failed).  Concurrent keys give the result of the
     *                 reference's sequential CPU loop (atomic updates)
     * range_count   : counts_focus[j] = min(2^32-1, sum of the global counts of the global leaves under focus leaf j)
     *                 for the num_idx leaf indices j listed in leaves_focus_idx (rangeCount, :279-301); leaves has
     *                 num_leaves + 1 keys
     * mark_macs     : markings[n] = 1 for every node outside the focus key range [focus_nodes[0],
     *                 focus_nodes[num_focus_nodes]) that fails the MAC (|min. distance|^2 < centers[n][3], centers =
     *                 Vec4<T>[num_nodes]) against one of the focus cells and, with limit_source, is not deeper than that
     *                 cell's level - 1 (markMacs, R/traversal/macs.hpp:199-270); markings is NOT cleared
     * count_sfc_gaps / fill_sfc_gaps : spanSfcRange (R/sfc/common.hpp:370-438) per pair of consecutive keys: the
     *                 number of nodes resp. the keys of the coarsest cornerstone sub-tree covering [tree[i], tree[i+1]);
     *                 fill writes them at new_tree + node_ops[i] (node_ops = exclusive scan of the counts, num_nodes + 1
     *                 entries) and the terminal key at new_tree[node_ops[num_nodes]]
     * geo_mac_spheres : spheres[n] = (geometric centre, (2 max(size) inv_theta)^2)   (computeMinMacR2)
     * set_mac       : spheres[n][3] <- (2 max(size) inv_theta + |spheres[n].xyz - geometric centre|)^2, 0 if it was 0
     *                 (computeVecMacR2 / setMac)
     * move_centers  : dst[n] = (src[n], 1) from Vec3<T> to Vec4<T>
     * leaf_source_centers : centre of |mass| of the particles of every leaf, stored at its node index
     *                 (computeLeafSourceCenterGpu; coord/mass/center bits = 64/64/64, 64/32/64 or 32/32/32)
     * upsweep_centers : centres of |mass| of the internal nodes, bottom-up (upsweepCentersGpu); level_range is a HOST
     *                 array of num_levels + 1 entries like in the reference
     * ------------------------------------------------------------------------------------------- */
    int cstone_hip_rebalance_decision_essential(cstone_hip_ctx* ctx, int key_bits, const void* prefixes,
                                                const int32_t* child_offsets, const int32_t* parents,
                                                const uint32_t* counts, const char* macs, uint64_t focus_start,
                                                uint64_t focus_end, uint32_t bucket_size, int32_t* node_ops,
                                                int num_nodes);
    int cstone_hip_mac_refine_decision(cstone_hip_ctx* ctx, int key_bits, const void* prefixes, const char* macs,
                                       const int32_t* leaf_to_internal, int num_leaves, int focus_first,
                                       int focus_last, int32_t* node_ops);
    int cstone_hip_protect_ancestors(cstone_hip_ctx* ctx, int key_bits, const void* prefixes, const int32_t* parents,
                                     int32_t* node_ops, int num_nodes, int* converged_host);
    int cstone_hip_enforce_keys(cstone_hip_ctx* ctx, int key_bits, const void* forced_keys, int num_forced_keys,
                                const void* prefixes, const int32_t* child_offsets, const int32_t* parents,
                                int32_t* node_ops, int* status_host);
    int cstone_hip_range_count(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves,
                               const uint32_t* counts, const void* leaves_focus, const int32_t* leaves_focus_idx,
                               int num_idx, uint32_t* counts_focus);
    int cstone_hip_mark_macs(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                             const int32_t* child_offsets, const void* centers, const cstone_box* box_host,
                             const void* focus_nodes, int num_focus_nodes, int limit_source, char* markings);
    int cstone_hip_count_sfc_gaps(cstone_hip_ctx* ctx, int key_bits, const void* tree, int num_nodes,
                                  int32_t* node_ops);
    int cstone_hip_fill_sfc_gaps(cstone_hip_ctx* ctx, int key_bits, const void* tree, int num_nodes,
                                 const int32_t* node_ops, void* new_tree);
    int cstone_hip_geo_mac_spheres(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                                   int num_nodes, void* spheres, float inv_theta, const cstone_box* box_host);
    int cstone_hip_set_mac(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                           int num_nodes, void* spheres, float inv_theta, const cstone_box* box_host);
    int cstone_hip_move_centers(cstone_hip_ctx* ctx, int real_bits, const void* src, int num_nodes, void* dst);
    /* FocusedOctree::addMacs (R/focus/octree_focus_mpi.hpp:601-610): halo_flags[i] = 1 for every leaf i whose node
     * (leaf_to_internal[i], the leaf part of the map) carries a MAC mark */
    int cstone_hip_add_macs(cstone_hip_ctx* ctx, const char* macs, const int32_t* leaf_to_internal, int num_leaves,
                            int32_t* halo_flags);
    int cstone_hip_leaf_source_centers(cstone_hip_ctx* ctx, int coord_bits, int mass_bits, int center_bits,
                                       const void* x, const void* y, const void* z, const void* m,
                                       const int32_t* leaf_to_internal, int num_leaves, const uint32_t* layout,
                                       void* centers);
    int cstone_hip_upsweep_centers(cstone_hip_ctx* ctx, int real_bits, int num_levels, const int32_t* level_range_host,
                                   const int32_t* child_offsets, void* centers);

    /* ---------------------------------------------------------------------------------------------
     * Building blocks of the locally essential tree on SEVERAL ranks: the device work behind the host state machine of
     * FocusedOctree (R/focus/octree_focus_mpi.hpp:108-273), its treelet exchanges (R/focus/exchange_focus.hpp:98-287) and
     * of Halos::computeLayout (R/halos/halos.hpp:205-222, R/domain/layout.hpp:91-190, R/domain/exchange_keys.hpp:63-119).
     * The state machine itself is plain host C++ on top of this ABI (cornerstone-octree_amd/csrc/let.hpp).
     * raise         : records `message` as the context's last error and returns `code` (host layers above the ABI
     *                 report their own failures through the same channel as the library's)
     * find_peers_mac : findPeersMac (R/traversal/peers.hpp:63-118): peer_flags_host[r] = 1 for every rank r whose SFC
     *                 range [assignment[r], assignment[r+1]) holds a leaf of the replicated global tree that fails the
     *                 mutual min-distance MAC (minVecMacMutual, R/traversal/macs.hpp:171-194) paired with a leaf inside
     *                 my_rank's range, found by the reference's dual traversal (R/traversal/traversal.hpp:135-188) from
     *                 the nodes that span my_rank's range; prefixes / child_offsets / level_range: the linked octree
     *                 of the global leaves (device), assignment_host: num_ranks + 1 keys (as uint64)
     * keys_missing  : flags[i] = 1 if keys[i] is not one of leaves[0 .. num_leaves] (the test of checkTreelets,
     *                 exchange_focus.hpp:104-115: k != leaves[findNodeAbove(leaves, num_leaves, k)]), else 0
     * partition_keys : with scan = exclusive scan of flags: keys with flag 1 go to set_out[scan[i]], the others to
     *                 unset_out[i - scan[i]] (either output may be NULL): pruneTreelets / the rejected keys (:118-171)
     * zero_ops_at_keys : node_ops[findNodeAbove(leaves, num_keys_in_leaves, keys[i])] = 0 (exchangeRejectedKeys, :186-190);
     *                 leaves has num_leaves + 1 keys, all of them are searched
     * locate_nodes  : idx[i] = locateNode(keys[i], keys[i+1], prefixes, level_range) for i < num_keys - 1
     *                 (indexTreelets, :266-287; R/tree/octree.hpp:216-241): the node with exactly that key range, or
     *                 num_nodes if there is none
     * node_layout   : computeNodeLayout (R/domain/layout.hpp:150-165): layout[num_leaves + 1] = exclusive scan of
     *                 (first <= i < last || flags[i]) ? counts[i] : 0
     * halo_requests : extractMarkedElements for every peer at once (R/domain/layout.hpp:104-139, as called by
     *                 exchangeRequestKeys, R/domain/exchange_keys.hpp:76-83): ranges_host = num_ranks pairs
     *                 {first leaf, last leaf} in ascending order ({0,0} for ranks that are no peers); for each run of
     *                 flagged leaves inside a peer's range the pair (leaves[run start], leaves[run end]) is written to
     *                 pairs_out (2 keys per run, peer after peer); pair_counts_host[r] = number of runs of rank r;
     *                 *unmatched_host = flagged leaves outside [first, last) that lie in no peer's range (checkHalos,
     *                 R/halos/halos.hpp:59-95).  pairs_out needs room for 2 * (flagged leaves) keys
     * ranges_from_keys : the serving side of exchangeRequestKeys (:98-108): range r = [layout[findNodeAbove(leaves,
     *                 pairs[2r])], layout[findNodeAbove(leaves, pairs[2r+1])]) -> range_offsets[r] = its start,
     *                 range_scan[num_pairs + 1] = exclusive scan of the lengths (what gather_ranges takes)
     * ------------------------------------------------------------------------------------------- */
    int cstone_hip_raise(cstone_hip_ctx* ctx, int code, const char* message);
    /* upload : a SMALL host array to the device without synchronising the stream: the bytes are copied to a pinned
     *          staging ring first, so src_host may be reused as soon as the call returns; ordered on the context's stream
     *          like any other work (arrays beyond a quarter of the ring, 256 KiB, go the way of memcpy_h2d) */
    int cstone_hip_upload(cstone_hip_ctx* ctx, void* dst, const void* src_host, size_t bytes);
    /* focus_update_ops : the decision part of CombinedUpdate::updateFocus (R/focus/octree_focus.hpp:97-122) in one call
     *          and ONE read-back: rebalance_decision_essential, enforce_keys (num_forced_keys device keys),
     *          protect_ancestors on node_ops_all[num_nodes], then the leaves' ops in leaf order (leaf_to_internal: the
     *          leaf part of the tree's map, num_leaves entries), scanned exclusively into leaf_ops[num_leaves + 1] (what
     *          rebalance_tree takes).  result_host = {status of the enforced keys, converged as updateFocus reports it,
     *          1 if every leaf keeps, new number of leaves} */
    int cstone_hip_focus_update_ops(cstone_hip_ctx* ctx, int key_bits, const void* prefixes,
                                    const int32_t* child_offsets, const int32_t* parents, const uint32_t* counts,
                                    const char* macs, uint64_t focus_start, uint64_t focus_end, uint32_t bucket_size,
                                    const void* forced_keys, int num_forced_keys, const int32_t* leaf_to_internal,
                                    int num_leaves, int num_nodes, int32_t* node_ops_all, int32_t* leaf_ops,
                                    int* result_host);
    int cstone_hip_find_peers_mac(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                                  const int32_t* child_offsets, const int32_t* level_range,
                                  const uint64_t* assignment_host, int num_ranks, int my_rank,
                                  const cstone_box* box_host, float inv_theta_eff, int32_t* peer_flags_host);
    int cstone_hip_keys_missing(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves, const void* keys,
                                size_t num_keys, uint32_t* flags);
    int cstone_hip_partition_keys(cstone_hip_ctx* ctx, int key_bits, const void* keys, const uint32_t* flags,
                                  const uint32_t* scan, size_t num_keys, void* set_out, void* unset_out);
    int cstone_hip_zero_ops_at_keys(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves,
                                    const void* keys, size_t num_keys, int32_t* node_ops);
    int cstone_hip_locate_nodes(cstone_hip_ctx* ctx, int key_bits, const void* keys, size_t num_keys,
                                const void* prefixes, const int32_t* level_range, int32_t* idx);
    int cstone_hip_node_layout(cstone_hip_ctx* ctx, const uint32_t* counts, const int32_t* flags, int first, int last,
                               int num_leaves, uint32_t* layout);
    int cstone_hip_halo_requests(cstone_hip_ctx* ctx, int key_bits, const void* leaves, const int32_t* flags,
                                 int num_leaves, int first, int last, const int32_t* ranges_host, int num_ranks,
                                 void* pairs_out, uint32_t* pair_counts_host, uint32_t* unmatched_host);
    /* halo_requests without the trip to the host: row_dev[p] = 2 * (key pairs requested from rank p) as u64 for p <
     * num_ranks, row_dev[num_ranks] = status word (2: external_failure != 0, 1: halo cells that belong to no peer, 0: fine)
     * -- the row this rank contributes to the all-gather of Halos::computeLayout, written where the collective reads it */
    int cstone_hip_halo_request_rows(cstone_hip_ctx* ctx, int key_bits, const void* leaves, const int32_t* flags,
                                     int num_leaves, int first, int last, const int32_t* ranges_host, int num_ranks,
                                     void* pairs_out, uint64_t* row_dev, int external_failure);
    /* translateAssignment (R/domain/domaindecomp.hpp:183-206) + the treelet sizes of syncTreelets
     * (R/focus/exchange_focus.hpp:61-96) on the device: bounds_dev[r] = first leaf >= assignment[r], bounds_dev[num_ranks +
     * 1 + r] = first leaf >= assignment[r] + 1 (r = 0 .. num_ranks: the output of cstone_hip_lower_bound for these 2
     * (num_ranks + 1) keys); row_dev[p] = leaves over rank p's range + 1 if is_peer_host[p], else 0 */
    int cstone_hip_peer_range_counts(cstone_hip_ctx* ctx, const uint64_t* bounds_dev, const uint8_t* is_peer_host,
                                     int num_ranks, uint64_t* row_dev);
    int cstone_hip_ranges_from_keys(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves,
                                    const uint32_t* layout, const void* pairs, size_t num_pairs,
                                    uint32_t* range_offsets, uint32_t* range_scan);

    /* ---------------------------------------------------------------------------------------------
     * neighbor search: replaces findNeighbors (R/findneighbors.hpp:160-188) / the traverseNeighbors
     * device function (R/traversal/find_neighbors.cuh:436-506) on an OctreeNsView
     * (R/tree/octree.hpp:297-317).  For i in [first,last): counts[i-first] = number of j != i with
     * |r_i - r_j|^2 < (2 h_i)^2 (minimum image if periodic); the first ngmax are stored row-major in
     * neighbors[(i-first)*ngmax + k] in the reference's CPU traversal order.
     * ------------------------------------------------------------------------------------------- */
    int cstone_hip_find_neighbors(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y, const void* z,
                                  const void* h, uint32_t first, uint32_t last, const cstone_box* box_host,
                                  const int32_t* child_offsets, const int32_t* internal_to_leaf,
                                  const uint32_t* layout, const void* centers, const void* sizes, float ext,
                                  uint32_t ngmax, uint32_t* neighbors, uint32_t* counts);

    /* the same search with the traversal counters of the reference's NcStats (R/traversal/find_neighbors.cuh:345-369,
     * 494-502; what its neighbor_driver prints, test/performance/neighbor_driver.cu:159-170):
     * stats_host[0] = sumP2P  : distance tests, summed over the targets (a target is tested against every particle of
     *                           every leaf its own walk reaches)
     * stats_host[1] = maxP2P  : most distance tests of one target
     * stats_host[2] = maxStack: deepest use of the traversal stack (entries; one stack per wave of 64 targets)
     * stats_host[3]           : tests the SIMDs issued: a wave tests every particle of a leaf in all 64 lanes, also for
     *                           the lanes whose own walk did not reach that leaf (>= sumP2P; their ratio is the lane
     *                           efficiency of sharing one traversal among 64 targets) */
    int cstone_hip_find_neighbors_stats(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y, const void* z,
                                        const void* h, uint32_t first, uint32_t last, const cstone_box* box_host,
                                        const int32_t* child_offsets, const int32_t* internal_to_leaf,
                                        const uint32_t* layout, const void* centers, const void* sizes, float ext,
                                        uint32_t ngmax, uint32_t* neighbors, uint32_t* counts, uint64_t* stats_host);

    /* ---------------------------------------------------------------------------------------------
     * target particle groups: replace computeFixedGroups (R/traversal/groups_gpu.h:46, groups_gpu.cu:41-71) and
     * computeGroupSplits (R/traversal/groups_gpu.h:73-87, groups_gpu.cu:74-151); the result is what a
     * GroupView (R/traversal/groups.hpp:20-26) points at: groupStart = groups, groupEnd = groups + 1.
     * fixed_groups : groups[g] = first + g * group_size for g < *num_groups = ceil((last-first)/group_size),
     *                groups[*num_groups] = last; groups holds *num_groups + 1 entries.
     * group_splits : runs of group_size (64 or 128 = one or two wavefronts; anything else is CSTONE_E_ARG where the
     *                reference throws) consecutive particles of [first,last), each cut again behind every particle
     *                whose successor in the run is farther away, in coordinates scaled by the box's inverse lengths,
     *                than tol_factor x 2^-level of the deepest leaf among the run's first 64 particles.  leaves /
     *                layout: cornerstone leaf keys [num_leaves + 1] and the index of each leaf's first particle.
     *                Writes *num_groups + 1 <= capacity ascending indices (first ... last); CSTONE_E_CAPACITY with
     *                *num_groups set when capacity is too small (last - first + 1 always suffices).  The smoothing
     *                lengths of the reference's signature do not enter the result and are not passed.
     * find_neighbors_groups : find_neighbors with the 64 targets of a wavefront taken from one group
     *                [group_start[g], group_end[g]) (clipped to [first,last); longer groups are walked 64 at a time) instead of 64
     *                consecutive indices; particles outside every group keep counts/neighbors untouched.  Same
     *                result rows as find_neighbors: counts[i-first], neighbors[(i-first)*ngmax + k].
     * ------------------------------------------------------------------------------------------- */
    int cstone_hip_compute_fixed_groups(cstone_hip_ctx* ctx, uint32_t first, uint32_t last, uint32_t group_size,
                                        uint32_t* groups, uint32_t* num_groups);
    int cstone_hip_compute_group_splits(cstone_hip_ctx* ctx, int key_bits, int real_bits, uint32_t first, uint32_t last,
                                        const void* x, const void* y, const void* z, const void* leaves,
                                        int num_leaves, const uint32_t* layout, const cstone_box* box_host,
                                        uint32_t group_size, float tol_factor, uint32_t* groups, size_t capacity,
                                        uint32_t* num_groups);
    /* the same search with the lists laid out like the warp-interleaved lists of traverseNeighbors
     * (R/traversal/find_neighbors.cuh:116, targetSize = 64): neighbour k of target t (counted from first) is
     * neighbors[((t / 64) * ngmax + k) * 64 + t % 64]; neighbors holds ceil((last - first) / 64) * 64 * ngmax entries */
    int cstone_hip_find_neighbors_interleaved(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y,
                                              const void* z, const void* h, uint32_t first, uint32_t last,
                                              const cstone_box* box_host, const int32_t* child_offsets,
                                              const int32_t* internal_to_leaf, const uint32_t* layout,
                                              const void* centers, const void* sizes, float ext, uint32_t ngmax,
                                              uint32_t* neighbors, uint32_t* counts);
    int cstone_hip_find_neighbors_groups(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y,
                                         const void* z, const void* h, uint32_t first, uint32_t last,
                                         const uint32_t* group_start, const uint32_t* group_end, uint32_t num_groups,
                                         const cstone_box* box_host, const int32_t* child_offsets,
                                         const int32_t* internal_to_leaf, const uint32_t* layout, const void* centers,
                                         const void* sizes, float ext, uint32_t ngmax, uint32_t* neighbors,
                                         uint32_t* counts);

    /* ---------------------------------------------------------------------------------------------
     * Domain: device-resident cstone::Domain<KeyType,T,GpuTag> (R/domain/domain.hpp:66-699).
     * create : Domain(rank, nRanks, bucketSize, bucketSizeFocus, theta, box) (:95-113); CSTONE_E_ARG if
     *          bucket_size < bucket_size_focus (the reference throws std::runtime_error). This round implements
     *          num_ranks == 1; other values are refused.
     * sync   : Domain::sync(keys, x, y, z, h, properties, scratch) (:196-243). Arrays of n elements (n = the
     *          previous num_particles_with_halos, or the initial particle count on the first call). Like the
     *          reference, which swaps the caller's vectors with the scratch vectors (layout.hpp:214-219), the call
     *          EXCHANGES buffers: on return *x, *y, *z, *h, props[i] and *scratch point to (possibly different)
     *          members of the set of buffers passed in; all must therefore have the same capacity of n elements of
     *          real_bits (properties: element size <= real_bits/8). *keys is sorted in place. Entries whose key slot
     *          holds the remove marker 2^(3 maxLevel) on entry are dropped (sfc.hpp:289).
     *          Post-conditions as domain.hpp:144-179: arrays SFC-sorted, keys consistent with x,y,z under box().
     * view   : accessors startIndex/endIndex/nParticlesWithHalos/box/globalTree/focusTree/layout and
     *          octreeProperties() (:388-437) as raw DEVICE pointers, valid until the next sync.
     * ------------------------------------------------------------------------------------------- */
    typedef struct cstone_hip_domain cstone_hip_domain;

    typedef struct cstone_hip_domain_view
    {
        uint32_t start_index, end_index, num_particles_with_halos; /* Domain::startIndex/endIndex/nParticlesWithHalos */
        cstone_box box;                                           /* Domain::box() */
        int32_t num_global_leaves;                                /* Domain::globalTree() */
        const void* global_leaves;                                /* K[num_global_leaves+1] */
        const uint32_t* global_counts;
        int32_t num_focus_leaves, num_focus_nodes; /* Domain::focusTree(), OctreeNsView (R/tree/octree.hpp:297-317) */
        const void* focus_leaves;                  /* K[L+1] */
        const uint32_t* focus_leaf_counts;         /* u32[L] */
        const void* prefixes;                      /* K[M] */
        const int32_t* child_offsets;
        const int32_t* parents;
        const int32_t* level_range;
        const int32_t* internal_to_leaf;
        const int32_t* leaf_to_internal;
        const uint32_t* layout; /* u32[L+1], Domain::layout() */
        const void* centers;    /* T[M][3] */
        const void* sizes;      /* T[M][3] */
        const int32_t* halo_flags;
        const uint32_t* sfc_order; /* the ordering kept in the last scratch buffer by the reference (:206) */
        const float* halo_radii;   /* f32[L]: 2 * haloSearchExt * max h per focus leaf (Halos::discover, halos.hpp:128-160) */
        const void* expansion_centers; /* T[M][4]: (centre of mass, MAC radius^2) per node after _sync_grav /
                                          _update_expansion_centers of THIS sync's tree, else NULL */
    } cstone_hip_domain_view;

    int cstone_hip_domain_create(cstone_hip_ctx* ctx, cstone_hip_domain** out, int curve, int key_bits, int real_bits,
                                 int rank, int num_ranks, uint32_t bucket_size, uint32_t bucket_size_focus, float theta,
                                 const cstone_box* box_host);
    int cstone_hip_domain_destroy(cstone_hip_domain* dom);
    int cstone_hip_domain_sync(cstone_hip_domain* dom, void** keys, void** x, void** y, void** z, void** h, size_t n,
                               void** scratch, void** props, const int* prop_bytes, int num_props);
    /* the same with num_scratch >= 1 scratch buffers of n elements each (scratch: array of num_scratch device
     * pointers), like the scratch TUPLE of the reference's sync (R/domain/domain.hpp:196-206).  From three buffers on
     * x, y and z are brought into SFC order by ONE kernel that reads the ordering once (cstone_hip_gather_multi: 52
     * instead of 60 bytes per particle for f64); with fewer the arrays rotate through scratch[0] one after the other.
     * From three buffers on that gather runs on a second stream of the context, next to the tree update; with FOUR, h
     * goes through the fourth while x, y, z are still on their way.  (CSTONE_FUSED_LEAF_PASS=1 with four buffers selects
     * the field-carrying leaf pass of the incremental re-sort instead -- x, y, z, h moved with the keys, 84 instead of
     * 92 bytes per particle behind the encode; measured slower, DESIGN.md 4c, kept for comparison.)
     * All buffers take part in the pointer exchange: on return *x, *y, *z, *h, props[i] and scratch[q] are a
     * permutation of the buffers passed in.  Same results whatever num_scratch is. */
    int cstone_hip_domain_sync_scratch(cstone_hip_domain* dom, void** keys, void** x, void** y, void** z, void** h,
                                       size_t n, void** scratch, int num_scratch, void** props, const int* prop_bytes,
                                       int num_props);
    /* Domain::syncGrav on ONE rank (R/domain/domain.hpp:246-325): with no peers the focus tree is the tree of sync (every
     * node is inside the focus: the MACs decide nothing), so this is _sync_scratch with the masses m (n values of
     * mass_bits = 32 | 64 bits; *m is exchanged like the other arrays, so its buffer offers n elements of the
     * coordinates' size like every property buffer) as one more property, followed by
     * _update_expansion_centers = Domain::updateExpansionCenters (:415-421): (centre of mass, MAC radius^2 for 1 / theta)
     * per node of the focus tree in view.expansion_centers (computeLeafSourceCenter, the CombineSourceCenter upsweep and
     * setMac of R/focus/source_center.hpp).  x, y, z, m of _update_expansion_centers: arrays laid out like the result
     * arrays of the last sync.  Several ranks: cstone_hip_domain_mr_sync_grav. */
    int cstone_hip_domain_sync_grav(cstone_hip_domain* dom, void** keys, void** x, void** y, void** z, void** h, void** m,
                                    int mass_bits, size_t n, void** scratch, int num_scratch, void** props,
                                    const int* prop_bytes, int num_props);
    int cstone_hip_domain_update_expansion_centers(cstone_hip_domain* dom, const void* x, const void* y, const void* z,
                                                   const void* m, int mass_bits);
    int cstone_hip_domain_view_get(cstone_hip_domain* dom, cstone_hip_domain_view* out);
    /* Domain::setHaloFactor (R/domain/domain.hpp:412): extra search factor of the halo discovery (default 1.0), lets a
     * client take several integration steps between syncs */
    int cstone_hip_domain_set_halo_factor(cstone_hip_domain* dom, float factor);
    /* How a sync orders the particles (results are identical in all modes): INCREMENTAL (default) repairs the order of
     * the previous sync leaf by leaf where it can, FROM_SCRATCH radix-sorts the digits above the previous
     * tree's leaf level and finishes the runs, ALL_DIGITS radix-sorts all key digits like the reference's GPU path.
     * set_speculative_box(0): the extents of an open box are always measured before the keys are computed.
     * The environment variables CSTONE_NO_RESORT / CSTONE_FULL_SORT / CSTONE_NO_SPECULATIVE_BOX override these. */
#define CSTONE_SORT_INCREMENTAL 0
#define CSTONE_SORT_FROM_SCRATCH 1
#define CSTONE_SORT_ALL_DIGITS 2
    int cstone_hip_domain_set_sort_mode(cstone_hip_domain* dom, int mode);
    int cstone_hip_domain_set_speculative_box(cstone_hip_domain* dom, int on);
    /* 1 when the library was built with -DCSTONE_TEST_HOOKS (lib/libcstone_hip_hooks.so: fault injection through
     * CSTONE_MR_FAIL_AT / CSTONE_FORCE_DEVICE_ERROR for the tests), 0 for the product build */
    int cstone_hip_test_hooks(void);
    /* Domain::reapplySync (R/domain/domain.hpp:334-378) on one rank: out[i] = in[sfc_order[i]] for the end_index
     * particles the last sync kept; in has the n elements (elem_bytes in {1,2,4,8,12,16,24,32}) of that call's arrays,
     * out must not alias in */
    int cstone_hip_domain_reapply_sync(cstone_hip_domain* dom, const void* in, size_t n, int elem_bytes, void* out);

    /* How the syncs of a (single-rank) domain went so far: which of them took the incremental re-sort (the sorted order
     * built from the previous sync's order, csrc/resort.hpp) instead of the radix sort of all keys, how often the
     * speculatively used box had changed.  Counters only; the results of a sync do not depend on the path taken. */
    typedef struct cstone_hip_domain_stats
    {
        uint32_t syncs;
        uint32_t resorts;             /* syncs ordered by the incremental re-sort */
        uint32_t resort_fallbacks;    /* re-sort attempts given up (too many movers, an overfull leaf) */
        uint32_t box_redos;           /* syncs whose box differed from the one the keys were first computed with */
        uint32_t full_sort_fallbacks; /* partial-digit sorts completed by a sort of the remaining digits */
        uint32_t last_movers;         /* particles that had left their leaf at the last re-sort */
    } cstone_hip_domain_stats;
    int cstone_hip_domain_stats_get(cstone_hip_domain* dom, cstone_hip_domain_stats* out);

    /* ---------------------------------------------------------------------------------------------
     * Domain::sync on SEVERAL ranks, one process per GPU (R/domain/domain.hpp:196-243 with the exchange steps of
     * GlobalAssignment R/domain/assignment.hpp:57-158, R/domain/domaindecomp_mpi.hpp:86-174 and of Halos
     * R/halos/halos.hpp:128-257).  The library does all the device work and the decomposition logic; the three
     * collectives it needs are provided by the host application through cstone_hip_comm_ops -- RCCL via
     * torch.distributed in this repository's bench (cstone_amd/distributed.py), MPI or plain RCCL elsewhere.
     * All buffers handed to the callbacks are DEVICE pointers on the context's device; the callbacks must have
     * completed (or be ordered on the context's stream) when they return; return 0 for success.
     * ------------------------------------------------------------------------------------------- */
    typedef struct cstone_hip_comm_ops
    {
        void* user;
        /* in-place reduction over all ranks; dtype 0 = f64, 1 = u32; op 0 = sum, 1 = min */
        int (*all_reduce)(void* user, void* buf, size_t count, int dtype, int op);
        /* every rank contributes `bytes` bytes; recv holds num_ranks * bytes, ordered by rank */
        int (*all_gather)(void* user, const void* send, void* recv, size_t bytes);
        /* segments for / from the ranks lie back to back in rank order; sizes in bytes (host arrays, num_ranks long) */
        int (*all_to_all_v)(void* user, const void* send, const size_t* send_bytes, void* recv,
                            const size_t* recv_bytes);
    } cstone_hip_comm_ops;

    /* cstone_hip_comm_ops served by RCCL inside the library (csrc/comm_rccl.hip): every collective is enqueued on the
     * context's stream -- no host synchronisation, no host-language callback.  all_reduce = ncclAllReduce, all_gather =
     * ncclAllGather, all_to_all_v = one group of ncclSend / ncclRecv per peer (xGMI links are point to point).
     * Bootstrap like any RCCL program: ONE rank obtains the 128-byte id (unique_id) and hands it to the others through
     * whatever channel the application has (a file, MPI_Bcast, a TCP store ...); then every rank calls create (collective,
     * blocks until all num_ranks ranks have joined) with the context of ITS GPU.  ops fills a cstone_hip_comm_ops that
     * stays valid until destroy.  librccl.so is opened on the first of these calls. */
    typedef struct cstone_hip_comm_rccl cstone_hip_comm_rccl;
    int cstone_hip_comm_rccl_unique_id(cstone_hip_ctx* ctx, void* id128_host);
    int cstone_hip_comm_rccl_create(cstone_hip_ctx* ctx, const void* id128_host, int rank, int num_ranks,
                                    cstone_hip_comm_rccl** out);
    int cstone_hip_comm_rccl_ops(cstone_hip_comm_rccl* comm, cstone_hip_comm_ops* ops);
    int cstone_hip_comm_rccl_destroy(cstone_hip_comm_rccl* comm);

    typedef struct cstone_hip_domain_mr cstone_hip_domain_mr;

    typedef struct cstone_hip_domain_mr_view
    {
        /* [halos of lower ranks | assigned, SFC sorted | halos of higher ranks]; arrays are owned by the domain and
         * stay valid until the next but one sync (two buffer sets alternate, so the client may update the assigned
         * range in place and pass it back as the next input) */
        uint32_t start_index, end_index, num_particles_with_halos, pad0_;
        cstone_box box;
        const void *keys, *x, *y, *z, *h;
        int32_t num_global_leaves, num_focus_leaves;
        const void* global_leaves;      /* K[num_global_leaves + 1], device */
        const uint32_t* global_counts;  /* device */
        const void* focus_leaves;       /* this rank's finest tree, device */
        const uint32_t* focus_leaf_counts;
        uint64_t range_start, range_end; /* the rank's SFC key range */
        uint64_t particles_sent, halos_received, halos_sent, halo_boxes_exported;
        const void* props[16];           /* the conserved properties of the last sync_props call, same layout; their
                                            halo ranges are NOT filled (exchange_halos does that on request) */
        uint64_t resorts;                /* syncs so far whose local particles were ordered by the incremental re-sort
                                            (csrc/resort.hpp) instead of the radix sort; same results either way */
        /* with CSTONE_MR_HALOS_LET (the default) focus_leaves / focus_leaf_counts are the reference's focusTree():
         * the locally essential tree over the WHOLE key range, and: */
        int32_t start_cell, end_cell;    /* Domain::startCell / endCell: this rank's leaves in focus_leaves */
        int32_t num_peers, pad1_;
        const uint32_t* layout;          /* Domain::layout(): u32[num_focus_leaves + 1], offsets into the result arrays */
        const int32_t* halo_flags;       /* i32[num_focus_leaves]: 1 for the leaves whose particles are here as halos */
    } cstone_hip_domain_mr_view;

    int cstone_hip_domain_mr_create(cstone_hip_ctx* ctx, cstone_hip_domain_mr** out, int curve, int key_bits,
                                    int real_bits, int rank, int num_ranks, uint32_t bucket_size,
                                    uint32_t bucket_size_focus, const cstone_box* box_host,
                                    const cstone_hip_comm_ops* comm);
    int cstone_hip_domain_mr_destroy(cstone_hip_domain_mr* dom);
    /* x, y, z, h: this rank's n particles in any order (device; may point into the previous result) */
    int cstone_hip_domain_mr_sync(cstone_hip_domain_mr* dom, const void* x, const void* y, const void* z,
                                  const void* h, size_t n);
    /* the same with further conserved per-particle fields (the `properties` of Domain::sync, R/domain/domain.hpp:196-203:
     * masses, velocities, ...; up to 16 arrays with elements of 1, 2, 4, 8, 12, 16, 24 or 32 bytes) that follow their particles to the new owner
     * and into SFC order */
    int cstone_hip_domain_mr_sync_props(cstone_hip_domain_mr* dom, const void* x, const void* y, const void* z,
                                        const void* h, size_t n, const void* const* props, const int* prop_bytes,
                                        int num_props);
    /* the same with the caller's key array (n keys, device, or NULL): entries that hold the remove marker
     * 2^(3*maxLevel) flag their particle for removal (R/sfc/sfc.hpp:284-291, R/tree/definitions.h:87-91), all other
     * entries are ignored */
    int cstone_hip_domain_mr_sync_keys(cstone_hip_domain_mr* dom, const void* keys, const void* x, const void* y,
                                       const void* z, const void* h, size_t n, const void* const* props,
                                       const int* prop_bytes, int num_props);
    /* Domain::syncGrav (R/domain/domain.hpp:246-325) on the RCCL route: like _sync_keys, with the masses m (n values of
     * mass_bits = 32 | 64 bits, device) following their particles as ONE MORE property behind the num_props given ones
     * (view.props[num_props] afterwards), and the focus tree resolved by the vector MAC on the mass centres of its nodes:
     * FocusedOctree::updateCenters with the global centre exchange, updateMacs, addMacs and the retry with a larger centre
     * drift tolerance when halo cells belong to nobody (:288-317), restated in csrc/let.hpp (FocusLet::updateGrav) and
     * compared with the reference's syncGrav under MPI (oracle/let_check.cpp, `grav`).  cstone_hip_domain_mr_octree_get
     * then also hands out the expansion centres.  Halo mode CSTONE_MR_HALOS_LET only. */
    int cstone_hip_domain_mr_sync_grav(cstone_hip_domain_mr* dom, const void* keys, const void* x, const void* y,
                                       const void* z, const void* h, const void* m, int mass_bits, size_t n,
                                       const void* const* props, const int* prop_bytes, int num_props);
    /* Domain::updateExpansionCenters (R/domain/domain.hpp:415-421): mass centres and MAC radii of the focus tree from the
     * particles as they are NOW; x, y, z, m: arrays laid out like the result arrays of the last sync (their assigned range
     * is read).  Collective. */
    int cstone_hip_domain_mr_update_expansion_centers(cstone_hip_domain_mr* dom, const void* x, const void* y,
                                                      const void* z, const void* m, int mass_bits);
    int cstone_hip_domain_mr_view_get(cstone_hip_domain_mr* dom, cstone_hip_domain_mr_view* out);
    /* Domain::exchangeHalos (R/domain/domain.hpp:381-386): repeats the halo exchange of the last sync for one more
     * field; array (device; elements of 1, 2, 4, 8, 12, 16, 24 or 32 bytes, e.g. Vec3<float>, Vec4<double>) is laid out
     * like the result arrays: its assigned range is read,
     * its halo ranges are overwritten with the owners' values */
    int cstone_hip_domain_mr_exchange_halos(cstone_hip_domain_mr* dom, void* array, int elem_bytes);
    /* Domain::reapplySync (R/domain/domain.hpp:334-378): moves one more per-particle field along the routes of the last
     * sync (same particles to the same ranks, same final slots), for fields that were not passed to sync itself.
     * in: device, n elements laid out like the INPUT arrays of the last sync (n must be that call's n, the reference's
     * checkSizesEqual); out: device, laid out like the result arrays (num_particles_with_halos elements of elem_bytes in
     * {1,2,4,8,12,16,24,32}); only the assigned range [start_index, end_index) is written.  Collective: every rank
     * calls it. */
    int cstone_hip_domain_mr_reapply_sync(cstone_hip_domain_mr* dom, const void* in, size_t n, int elem_bytes,
                                          void* out);
    int cstone_hip_domain_mr_set_halo_factor(cstone_hip_domain_mr* dom, float factor);
    /* How the halos of a multi-rank sync are found; to be chosen before the first sync.
     * CSTONE_MR_HALOS_LET (default): the reference's way, R/domain/domain.hpp:217-237 -- peers from the MAC on the global
     *   tree, the locally essential (focus) tree with treelet / count exchanges between peers, halo discovery on it and
     *   key-range requests to the owners (csrc/let.hpp).  Focus tree, layout(), nParticlesWithHalos() and the halo
     *   particles are the reference's, bit for bit.
     * CSTONE_MR_HALOS_OWNER_SIDE: every rank exports the dilated boxes of its boundary leaves, the owners answer on
     *   their own finest trees: fewer exchange steps, the halo set is complete but may differ from
     *   the reference's by the particles of a few cells; focus_leaves is then this rank's own finest tree. */
#define CSTONE_MR_HALOS_LET 0
#define CSTONE_MR_HALOS_OWNER_SIDE 1
    int cstone_hip_domain_mr_set_halo_mode(cstone_hip_domain_mr* dom, int mode);
    /* the opening angle theta of the Domain constructor (R/domain/domain.hpp:95-113), default 0.5; before the first sync */
    int cstone_hip_domain_mr_set_theta(cstone_hip_domain_mr* dom, float theta);
    int cstone_hip_domain_mr_set_sort_mode(cstone_hip_domain_mr* dom, int mode);

    /* Domain::octreeProperties() and Domain::layout() (R/domain/domain.hpp:388-437) for the result arrays of the last
     * sync: a cornerstone tree with bucket_size_focus over ALL local particles, halos included, as an OctreeNsView
     * (R/tree/octree.hpp:297-317) -- what cstone_hip_find_neighbors / find_neighbors_groups / compute_group_splits
     * take.  layout[i] = index in the result arrays of the first particle of leaf i.  Built on the first request after
     * a sync (device pointers, valid until the next sync); local call, no collective. */
    typedef struct cstone_hip_domain_mr_octree
    {
        int32_t num_leaves, num_nodes;
        const void* leaves;          /* K[num_leaves + 1] */
        const uint32_t* leaf_counts; /* u32[num_leaves] */
        const void* prefixes;        /* K[num_nodes] */
        const int32_t* child_offsets;
        const int32_t* parents;
        const int32_t* level_range;
        const int32_t* internal_to_leaf;
        const int32_t* leaf_to_internal;
        const uint32_t* layout; /* u32[num_leaves + 1] */
        const void* centers;    /* T[num_nodes][3] */
        const void* sizes;      /* T[num_nodes][3] */
        const void* expansion_centers; /* T[num_nodes][4]: (centre of mass, MAC radius^2) after _sync_grav /
                                          _update_expansion_centers (FocusedOctree::expansionCenters), else NULL */
    } cstone_hip_domain_mr_octree;
    int cstone_hip_domain_mr_octree_get(cstone_hip_domain_mr* dom, cstone_hip_domain_mr_octree* out);

#ifdef __cplusplus
}
#endif
#endif /* CSTONE_HIP_H */
