// ticket_cut.hpp -- how queued tickets of one shape leave as device batches (session.hpp: ticket_dispatch).
//
// What is queued, `queued` transactions, leaves in round(queued / target) device batches of EQUAL size (to the ticket): a run
// of any length then pays for the one tail behind its last batch and never for a short straggler batch as well (VERDICT r05
// weak 4: with "fill up to the target" the headline peaked where the step count was a multiple of the target).  The cut is
// decided once and every part of it leaves in the same call: parts still `owed` take ceil(queued / owed) each.
// No HIP here: the same two functions are driven by the CPU tests through libzkhost (zkhost_ticket_cut).
#pragma once
#include <cstddef>
#include <vector>

namespace zk {

inline size_t ticket_parts(size_t queued, size_t target) {
  if (target == 0) return 1;
  const size_t p = (queued + target / 2) / target;
  return p ? p : 1;
}
inline size_t ticket_quota(size_t queued, size_t owed) { return owed ? (queued + owed - 1) / owed : queued; }

// the batches a burst of tickets (sizes in queue order) leaves in when everything is dispatched at once: tickets per batch
inline std::vector<size_t> ticket_cut(const std::vector<size_t>& sizes, size_t target) {
  std::vector<size_t> out;
  size_t queued = 0;
  for (size_t s : sizes) queued += s;
  size_t at = 0, owed = ticket_parts(queued, target);
  while (at < sizes.size()) {
    if (owed == 0) owed = ticket_parts(queued, target);
    const size_t quota = ticket_quota(queued, owed);
    size_t total = 0, n = 0;
    while (at < sizes.size()) {
      total += sizes[at++]; ++n;
      if (total >= quota) break;
    }
    out.push_back(n);
    queued -= total;
    --owed;
  }
  return out;
}

}  // namespace zk
