"""CPU oracle for the self-supervised anomaly-detection hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product path (``self-supervised-anomaly-detection_amd/``) never imports this
package and fails loudly when its HIP extension is missing.

What it is: a torch-CPU / numpy fp32 restatement of the reference's algorithm
for the path named in BASELINE.json (PeraNet forward / train step, cosine 3-NN
scoring, blur + bilinear upsample, cut-paste primitives).  Each function cites
the reference file:line it follows (paths relative to /root/reference).

Pinning: the reference has no golden vectors or known-answer tests of its own
(SURVEY.md section 4, F9).  The oracle is pinned against outputs of the
reference itself, run in the build container through throw-away third-party
stubs (tests/golden/make_fixtures.py); the resulting vectors are committed
under tests/golden/ and tests/test_oracle_golden.py replays them.  Arithmetic
that lives in un-vendored third-party packages (torchvision resnet18 /
gaussian_blur; no versions pinned by the reference) is restated from the public
spec and flagged "restated, third-party" where it occurs.
"""

from . import resnet18, peranet, scoring, weights  # noqa: F401
