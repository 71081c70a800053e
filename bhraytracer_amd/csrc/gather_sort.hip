// gather_sort.hip — the cell sort of the photon gather's queries as a library radix sort (round 3).
// The gather walks its queries in Morton-cell order (kernels.hip::RunGather).  Until round 3 that order came from a counting sort over the 2^27
// cells: one atomic per incoherent query into a 512 MB table, twice (41.8 ms per C5 frame), then once (33 ms).  A stable LSD radix sort of
// (cell, query) pairs streams instead — 16.4 ms for the 5.8e8 pairs of a C5 frame on MI355X (rocPRIM behind hipCUB: a plain library sort is what
// the guide asks for where nothing has to be fused) — and leaves the queries of a cell in index order.  Own translation unit: the header costs 10 s
// of compile time that kernels.hip does not have to pay.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

namespace bhrt {

// temp == nullptr: *temp_bytes = what the sort needs; otherwise sorts n pairs by the low `end_bit` bits of the key.  Returns a hipError_t.
int GatherSortPairs(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n, void *temp, size_t *temp_bytes, int end_bit,
                    hipStream_t stream)
{
    return (int)hipcub::DeviceRadixSort::SortPairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, stream);
}

} // namespace bhrt
