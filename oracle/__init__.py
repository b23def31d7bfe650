"""CPU oracle for the SlowFast training hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain ``torch.nn`` (fp32, CPU) restatement of the arithmetic the reference runs
through ``train.py`` / ``model/my_slowfast.py``.  It is the checker the HIP path is compared with.

Rules ("Oracle"):
  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
    import it; the product package ``video-classification_amd`` never does;
  * it never reads ``/root/reference`` at run time (the reference does not travel to the GPU box).

Pinning status:
  * ``FuseFastToSlow`` (reference-own arithmetic, model/my_slowfast.py:260-344) is pinned bit-exactly by
    ``tests/golden/fuse_fast_to_slow_*.npz`` captured from the reference class itself
    (``tests/golden/make_fuse_golden.py``).
  * the backbone lives in third-party ``pytorchvideo`` (un-vendored, un-pinned, not installed; SURVEY.md
    section 8c).  Its wiring is restated from SURVEY.md appendix A1 and pinned only structurally
    (parameter counts 34,566,488 / 38,077,321, conv MACs 65.709 G @256^2, the state-dict key set incl. the 12
    keys of train.py:94-108).  Numerically that part is "parity unpinned".
"""
