// seed_kernels.hip -- batched maximal-exact-match seeding on gfx950.
//
// Replaces, for a whole batch of reads at once, the reference's per-read call chain
//   IdentifySeedPairs_FastMode / _SensitiveMode  (src/AlignmentCandidates.cpp:49-80, 132-169)
//     -> BWT_Search                              (src/bwt_search.cpp:140-184)
//          -> bwt_2occ4 / bwt_occ4               (src/bwt_search.cpp:68-118)
//          -> bwt_sa -> bwt_invPsi -> bwt_occ    (src/bwt_search.cpp:44-66, 120-138)
//   + the final std::sort by (PosDiff,rPos) / (gPos,rPos).
//
// Three kernels:
//   search_kernel : persistent lanes, one read per lane.  A read is a chain of dependent
//                   searches (the next start depends on the previous match length), a search is a
//                   chain of dependent LF steps; each loop iteration performs at most one LF step
//                   per lane = one or two 64-byte Occ-block gathers.  Lanes whose read is
//                   exhausted pull the next read from a global queue with a wave-aggregated
//                   atomic (ballot + mbcnt), so a wave keeps its 64 intervals live until the batch
//                   is drained.  Only the reverse-complement side of BWA's bi-interval is tracked:
//                   on a forward+revcomp text it is a plain backward search and needs ONE base's
//                   rank at two positions per step instead of all four (SURVEY.md App. C check).
//                   When the interval has shrunk to a single suffix (and the full SA is resident) the
//                   lane stops ranking: one gather from the SA names the text position, and the
//                   rest of the match is a 48-bases-per-iteration comparison of the read against
//                   the 2-bit text -- same match length by construction (an interval of one extends
//                   iff the next text base equals the next read base), ~4x fewer cache lines.
//                   Output: a dense list of hits {interval start, size, rPos, len, read, seed slot}.
//   locate_kernel : persistent lanes over (hit, i) items.  SAMPLED mode walks LF until a sampled
//                   rank (bwt_sa); FULL mode is one gather from the expanded suffix array.  The
//                   text position of the pattern is 2L - SA[x1+i] - len.
//   sort_*_kernel : per-read ordering with the mode's comparator (total order, so the result is
//                   unique and equals the reference's std::sort output).
// The kernels live in topic fragments under kernels/ that are included below into this ONE translation unit.
#include "seed_kernels.hpp"

#include <hipcub/hipcub.hpp>
#include <cstdlib>

namespace kg {

#include "kernels/device_util.inc"
#include "kernels/search.inc"
#include "kernels/locate.inc"
#include "kernels/sort.inc"
#include "kernels/chain.inc"
#include "kernels/index_build.inc"
#include "kernels/launch.inc"

}  // namespace kg
