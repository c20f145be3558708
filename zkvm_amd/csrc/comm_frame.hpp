// comm_frame.hpp -- the framing of the one exchange of the sharded verification: what a rank
// contributes to the all-gather of the accept bitmaps, and how the gathered slots become the bitmap of the whole batch.
// Pure host code, shared by the product (session.hpp, zkgpu_comm_allgather_bitmap) and by the CPU test tier
// (hostlib.cpp): the transport -- ncclAllGather over xGMI -- moves `world` slots of equal size and knows nothing of this.
//
//   slot  := status:i32 | pad:4 | bitmap[width]          width = ceil(max shard / 8) rounded up to 8 bytes
//   rank r holds transactions [cuts[r], cuts[r+1]) of the batch; bit j of its bitmap = transaction cuts[r] + j
//
// Fail-closed across ranks: a non-zero status word in ANY slot (a rank's own error code, or the poison word a rank's
// send buffer holds when its copy to the device failed) gives every rank an error and an all-zero bitmap.
#pragma once
#include <cstdint>
#include <cstring>

namespace zk {
namespace commframe {

constexpr int OK = 0, EINVAL_ = -1, EREMOTE_ = -7;      // ZKGPU_OK / ZKGPU_EINVAL / ZKGPU_EREMOTE (include/zkgpu.h)

// bytes per slot for these cuts; 0 when the cuts are not non-decreasing
inline size_t slot_bytes(const uint64_t* cuts, int world) {
  size_t width = 0;
  for (int r = 0; r < world; ++r) {
    if (cuts[r + 1] < cuts[r]) return 0;
    const size_t w = (size_t)((cuts[r + 1] - cuts[r] + 7) / 8);
    if (w > width) width = w;
  }
  return 8 + ((width + 7) & ~(size_t)7);
}

// this rank's slot (slot bytes at `out`, zero-filled first).  A missing bitmap with an OK status is itself a fault and
// travels as one.
inline void pack(uint8_t* out, size_t slot, const uint64_t* cuts, int rank, const uint8_t* local_bitmap, int local_status) {
  memset(out, 0, slot);
  const uint64_t mine_n = cuts[rank + 1] - cuts[rank];
  const int32_t st = (local_status == OK && mine_n && !local_bitmap) ? EINVAL_ : (int32_t)local_status;
  memcpy(out, &st, 4);
  if (st == OK && mine_n) memcpy(out + 8, local_bitmap, (size_t)((mine_n + 7) / 8));
}

// the gathered slots (rank-major) -> the bitmap of the whole batch ((cuts[world] + 7) / 8 bytes at `whole`).
// Returns OK, this rank's own status, or EREMOTE_ when another rank reported one; `whole` is all zero unless OK.
inline int unpack(const uint8_t* all, size_t slot, const uint64_t* cuts, int world, int rank, uint8_t* whole) {
  const uint64_t batch = cuts[world];
  memset(whole, 0, (size_t)((batch + 7) / 8));
  int rc = OK;
  for (int r = 0; r < world; ++r) {
    int32_t s;
    memcpy(&s, all + slot * (size_t)r, 4);
    if (s != OK && rc == OK) rc = (r == rank) ? (int)s : EREMOTE_;
  }
  if (rc != OK) return rc;
  for (int r = 0; r < world; ++r) {
    const uint8_t* bmr = all + slot * (size_t)r + 8;
    for (uint64_t j = 0, i = cuts[r]; i < cuts[r + 1]; ++j, ++i)
      if ((bmr[j / 8] >> (j % 8)) & 1) whole[i / 8] |= (uint8_t)(1u << (i % 8));
  }
  return OK;
}

}  // namespace commframe
}  // namespace zk
