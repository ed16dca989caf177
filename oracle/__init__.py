"""CPU oracle for the barc4dip signal/metrics hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: it is a
plain NumPy/SciPy restatement of the reference algorithms (each function cites
the reference ``file:line`` it follows, paths relative to
``/root/reference/src/barc4dip``).  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the
checker -- never as the thing shipped or measured as the product.

Parity status: PINNED for everything whose arithmetic is NumPy/SciPy -- the
restatement is compared with outputs of the real reference (imported in the
build container by ``oracle/make_golden.py``; vectors committed under
``tests/golden``).  UNPINNED for the scikit-image / OpenCV back-ends
(``template_matching``, ``deconvolve_psf``): those libraries are absent from the
image, so ``oracle.wiener`` follows the published Wiener-Hunt algorithm and is
self-checked only.
"""
from . import signal_np, metrics_np, temporal_np, wiener_np  # noqa: F401
