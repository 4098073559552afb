"""``WCRBFNet.bind`` skips an upload only when the leaves are provably the ones it uploaded: NumPy
leaves are compared by content, torch leaves by object identity (strong reference held) + data_ptr + version.  The
fingerprint logic is host-only, so it is tested here without a GPU."""
import gc

import numpy as np
import torch

from irbfn_amd.model import WCRBFNet

fp_of = lambda leaves: WCRBFNet._fingerprint(leaves, torch)
same = WCRBFNet._same_fingerprint


def test_fresh_readonly_temporaries_with_new_contents_are_never_mistaken_for_the_cached_ones():
    """create -> freeze -> bind -> drop, every step with new contents: ids and buffers get recycled (the advisor saw
    identical (id, ptr, shape) keys in 18 of 19 steps); the content fingerprint differs every time."""
    cached, recycled = None, 0
    seen_ids = set()
    for step in range(40):
        leaves = [np.full((4, 7), float(step), np.float32), np.full((4,), float(step), np.float32)]
        for a in leaves:
            a.setflags(write=False)
        recycled += int(id(leaves[0]) in seen_ids)
        seen_ids.add(id(leaves[0]))
        fp = fp_of(leaves)
        assert not same(cached, fp), step
        cached = fp
        del leaves, fp
        gc.collect()
    # identical contents in a fresh array: provably the same parameters -> the upload may be skipped
    again = [np.full((4, 7), 39.0, np.float32), np.full((4,), 39.0, np.float32)]
    assert same(cached, fp_of(again))


def test_thaw_mutate_refreeze_is_seen():
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    a.setflags(write=False)
    f0 = fp_of([a])
    a.setflags(write=True)
    a[0, 0] = 99.0
    a.setflags(write=False)
    assert not same(f0, fp_of([a]))
    assert not same(f0, fp_of([a.astype(np.float64)]))          # dtype is part of the key
    assert not same(f0, fp_of([a.reshape(4, 3)]))


def test_torch_leaves_identity_is_held_by_a_strong_reference():
    t = torch.zeros(8)
    f0 = fp_of([t])
    assert same(f0, fp_of([t]))
    t.add_(1.0)                                                  # in-place write bumps _version
    assert not same(f0, fp_of([t]))
    f1 = fp_of([t])
    tid = id(t)
    del t
    gc.collect()
    # the cached fingerprint still references the tensor: a new tensor cannot take over its id while it is cached
    for _ in range(64):
        u = torch.zeros(8)
        assert id(u) != tid and not same(f1, fp_of([u]))
    assert not same(f1, fp_of([np.zeros(8, np.float32)]))        # kinds never compare equal
    assert not same(None, f1) and not same(f1, f1[:0])
