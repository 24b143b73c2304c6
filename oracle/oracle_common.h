/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the reference executor's algorithms for the
 * SpMV + Krylov hot path of Ginkgo 1.5.0.  Each function cites the
 * reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (repo-8852-ginkgo_amd/) never does.
 *
 * Pinning: the reference needs cmake-generated code (ginkgo/config.hpp from
 * include/ginkgo/config.hpp.in) and is therefore not built here.  The oracle
 * is pinned against the known-answer vectors of the reference's own tests
 * (reference/test/...), transcribed as data into tests/golden/ JSON files, against the reference's
 * matrices/test/ MatrixMarket data and examples/simple-solver/doc/results.dox -- see
 * tests/test_oracle_golden.py.
 *
 * Floating point: compiled with -ffp-contract=off, matching the reference
 * built for baseline x86-64 (no FMA): `c += val * b` is a rounded product
 * followed by a rounded sum.
 */
#ifndef GKO_ORACLE_COMMON_H_
#define GKO_ORACLE_COMMON_H_

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t i32;
typedef int64_t i64;
typedef uint8_t u8;
typedef uint64_t u64;

#define ORACLE_API __attribute__((visibility("default")))

/* stopping_status (include/ginkgo/core/stop/stopping_status.hpp:144-147) */
#define ST_CONVERGED 0x80u
#define ST_FINALIZED 0x40u
#define ST_ID_MASK 0x3fu

static inline int st_has_stopped(u8 s) { return (s & ST_ID_MASK) != 0; }
static inline u8 st_converge(u8 s, u8 id, int fin)
{
    if (!st_has_stopped(s)) {
        s |= ST_CONVERGED | (id & ST_ID_MASK);
        if (fin) s |= ST_FINALIZED;
    }
    return s;
}
static inline u8 st_stop(u8 s, u8 id, int fin)
{
    if (!st_has_stopped(s)) {
        s |= (id & ST_ID_MASK);
        if (fin) s |= ST_FINALIZED;
    }
    return s;
}

#endif
