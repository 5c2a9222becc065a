"""CPU oracle for the online parameterized-QG stepping path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pyqg_generative_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker / the timed CPU baseline.

Pinning status (see DESIGN.md "Oracle"):
  * generator half (``gen_ref``, ``samplers_ref``, ``operators_ref``,
    ``spectral_ref``): pinned by golden vectors produced by importing the
    reference's own Python (tests/golden/make_golden.py, committed .npz files).
  * spectral half (``qg_ref``): restatement of pyqg 0.7.2, a third-party
    dependency that is absent from /root/reference and not installable here:
    **parity unpinned** by reference outputs; guarded by the conservation /
    identity invariants the reference's notebooks check.
"""
