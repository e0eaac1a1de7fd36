// k_sort.hip -- descending sort of 64-bit corner keys (response key << 32 | y << 16 | x).
//
// This is the std::sort(tmpCorners, greaterThanPtr()) step of cv2.goodFeaturesToTrack
// (s1_lucaskanade_tracking.py:437), applied only to the ACCEPTED corners (see
// k_corners.hip).  It is the one place the library uses a ROCm header-only primitive
// (rocPRIM device radix sort, compiled into libicelk.so; no runtime dependency): a <=10^5-key sort
// once per detection frame is not on the per-pair critical path.  Kept in its own translation unit
// so a hand-written segmented sort can replace it without touching the detector.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "icelk_internal.h"

namespace icelk {

size_t sort_tmp_bytes(int n)
{
    size_t bytes = 0;
    unsigned long long* p = nullptr;
    (void)rocprim::radix_sort_keys_desc(nullptr, bytes, p, p, (size_t)(n > 0 ? n : 1), 0, 64, (hipStream_t)0);
    return bytes;
}

void sort_keys_desc(hipStream_t s, DetectScratch& D, const unsigned long long* in, unsigned long long* out, int n)
{
    if (n <= 0) return;
    size_t bytes = D.sort_tmp_bytes;
    (void)rocprim::radix_sort_keys_desc(D.sort_tmp, bytes, in, out, (size_t)n, 0, 64, s);
}

// ascending sort of (cell << 32 | point index) keys for the gridding step (k_grid.hip); temporary storage is the
// caller's (queried with tmp == nullptr)
size_t sort_keys_asc(hipStream_t s, void* tmp, size_t tmp_bytes, const unsigned long long* in, unsigned long long* out,
                     int n, int end_bit)
{
    size_t bytes = tmp_bytes;
    (void)rocprim::radix_sort_keys(tmp, bytes, in, out, (size_t)(n > 0 ? n : 1), 0, (unsigned)end_bit, s);
    return bytes;
}

}  // namespace icelk
