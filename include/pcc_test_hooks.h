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

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PCC_TEST_HOOKS_H */
