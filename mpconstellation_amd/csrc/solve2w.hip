// solve2w.hip -- the small-batch build of the solver: solve.hip compiled a second time, in its own namespace, with two waves
// per satellite (see the head of solve.hip and riccati_factor2 there).  Up to 512 satellites -- one wave per SIMD of the
// chip or fewer -- a satellite's time is the dependent chain of one wave; the second wave takes half of the factorisation's
// chain.  Replaces nothing else: the C entry points live in solve.hip and call mpcx2w_launch for such batches.
#define MPCX_TWO_WAVE 1
#include "solve.hip"
