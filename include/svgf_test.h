/* svgf_test.h — test hooks of libsvgf_mi355x.so's strip driver: transports that put every rank of a partition into one process on one
 * device, fault injection into their matching, and the message schedule as pure geometry.  Not for hosts. */
#ifndef SVGF_MI355X_TEST_H
#define SVGF_MI355X_TEST_H

#include "svgf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Further values of svgf_strips_create's `transport`:
 *   SVGF_TRANSPORT_RCCL_LOOPBACK  tests and the one-GPU simulation: ONE communicator of size 1 (comms[0]); every peer is its rank 0 and all virtual
 *                                 ranks share one communication stream.  Exercises RCCL's groups and kernels, not the peer addressing.
 *   SVGF_TRANSPORT_MAILBOX        every rank of the partition lives in this process (nlocal == world, `comms` ignored) with a communication
 *                                 stream of its own, addresses its neighbours by their real rank numbers — the code path of a multi-GPU run — and the
 *                                 library matches each send to the receive its peer posted for it (posting order per {source, destination} pair,
 *                                 same group: RCCL's rule) and turns the pair into a device-to-device copy on the receiver's stream.  A send nobody
 *                                 receives, a receive nobody sends or a size mismatch — what deadlocks a real run — fails the frame with
 *                                 SVGF_ERR_COMM.  Not a product transport: it cannot cross a process boundary. */
enum svgf_test_transport { SVGF_TRANSPORT_RCCL_LOOPBACK = 1, SVGF_TRANSPORT_MAILBOX = 2 };
/* The messages of ONE frame as rank `rank` posts them, in posting order (pure geometry, no device): what svgf_strips_frame hands to the
 * transport.  exchange 0 = the frame's state for the next frame's reprojection (posted once iteration 0 has fed the colour back,
 * waited for at the start of the next frame); exchange g >= 1 = the filter rows in front of iteration group g of the halo plan.
 * Every send has its mirror among the peer's receives of the same exchange — same plane, same global rows, same bytes — in the same
 * order per pair of ranks (tests/test_strips_cpu.py walks world = 2..8).  *count receives the number of messages; SVGF_ERR_INVALID
 * if it exceeds `capacity` (the first `capacity` are written). */
typedef struct svgf_strip_message {
    int exchange;
    int send;                 /* 1: this rank sends, 0: it receives */
    int peer;                 /* the neighbour's rank */
    int plane;                /* svgf_plane */
    int row_begin, row_end;   /* global rows */
    size_t bytes;
} svgf_strip_message;
int svgf_strips_messages(int width, int height, int rank, int world, int steps, int plan, int moments_radius, int motion_reach, int storage,
                         svgf_strip_message* out, int capacity, int* count);
/* SVGF_TRANSPORT_MAILBOX only, for the tests of the matching itself: the next send / receive rank `rank` posts is dropped, or its next
 * receive posted with half its size — the defects of a schedule that a multi-GPU run would answer with a hang.  The frame that meets the
 * defect fails with SVGF_ERR_COMM (the text names the ranks and the bytes) and the driver refuses further frames. */
enum svgf_mailbox_fault { SVGF_FAULT_NONE = 0, SVGF_FAULT_DROP_SEND = 1, SVGF_FAULT_DROP_RECV = 2, SVGF_FAULT_SHORT_RECV = 3 };
int svgf_strips_mailbox_fault(svgf_strips* s, int rank, int fault);
/* SVGF_TRANSPORT_MAILBOX only: groups matched, copies enqueued and bytes copied so far (any pointer may be NULL). */
int svgf_strips_transport_stats(const svgf_strips* s, unsigned long long* groups, unsigned long long* copies, unsigned long long* bytes);

#ifdef __cplusplus
}
#endif
#endif /* SVGF_MI355X_TEST_H */
