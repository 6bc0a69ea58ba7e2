/*
 * pcc_test_hooks.h -- TEST-ONLY entry points of libpcc_structural.so.  Not part of the product ABI
 * (pcc_structural.h / pcc_neighbour.h / pcc_emd.h): every function here is inert and returns 0 unless the
 * environment held PCC_TEST_HOOKS=1 when it was first called (tests/conftest.py sets it).
 */
#ifndef PCC_TEST_HOOKS_H
#define PCC_TEST_HOOKS_H

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* Makes the next cluster launch of pcc_auction_forward on the current device start with its error word raised, as if
 * a sample barrier had timed out: exercises the failure-reporting path of pcc_emd.h.  Returns 1 when armed. */
int pcc_test_inject_auction_failure(void);

/* A/B switches of the measurement scripts under tools/ and of the bit-identity tests (value 0 = the product's
 * behaviour).  They replace the PCC_* environment variables earlier rounds read inside the product library: nothing
 * in the product path selects behaviour from the environment any more.  Returns 1 when armed. */
enum {
    PCC_TUNE_PAIR_PLAIN_ORDER = 1, /* am_pair_kernel: row tiles in index order instead of near-pairs-first */
    PCC_TUNE_AM_NOCULL = 2,        /* approxmatch: every exact-zero skip off (the no-skip roofline of bench.py) */
    PCC_TUNE_AM_NOSPLIT = 3,       /* approxmatch: everything on the caller's stream (no half-batch lanes) */
    PCC_TUNE_AM_NORESIDENT = 4,    /* approxmatch: levels 0-2 as one launch per pass even where the resident launch qualifies */
    PCC_TUNE_EDGE_SCATTER = 5,     /* gather / edge-feature backward: the per-edge ds_add_f32 scatter */
    PCC_TUNE_NBRSUM_SCATTER = 6,   /* neighbour-sum backward: the per-edge ds_add_f32 scatter */
    PCC_TUNE_AUCTION_CLUSTER = 7,  /* auction: value 1 = one workgroup per sample, 2..16 = that many per sample */
    PCC_TUNE_KNN_NOSPLIT = 8,      /* c >= 4 k-NN: 1 = the 128-query kernel everywhere, 2 = the role-split kernel everywhere */
    PCC_TUNE_NN_HEAD = 11,         /* pcc_chamfer_emd: every lane searches its nearest neighbours right behind its sort */
    PCC_TUNE_AM_LANES = 9          /* approxmatch: number of half-batch lanes (1..4) instead of 2 */
};
int pcc_test_set_tuning(int key, int value);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PCC_TEST_HOOKS_H */
