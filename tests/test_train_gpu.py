"""GPU: one SFA training step (BASELINE.json configs[4]) against numpy / scipy in float64.
Tolerance (BASELINE.json): eigen-quantities to 1e-5."""
import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


def test_sfa_train_layer_matches_scipy(native_lib):
    from pyfaceanalysis_amd import nodes as N, synth
    from pyfaceanalysis_amd.train import sfa_train_layer
    n, side = 3000, 32
    seq = synth.make_training_sequence(n, side, seed=11)                  # (n, 1024) integer pixels, time ordered
    sb = N.Rectangular2dSwitchboard((side, side), (4, 4), (4, 4), 1)
    conn = sb.connections.reshape(-1, 16)                                 # 64 nodes x 16 inputs
    for dt in (np.uint8, np.float32):
        evals, evecs, mean, tms = sfa_train_layer(seq.astype(dt), conn)
        assert evals.shape == (64, 16) and evecs.shape == (64, 16, 16)
        for k in (0, 7, 33, 63):
            xk = seq[:, conn[k]]
            B = np.cov(xk.T)
            dx = xk[1:] - xk[:-1]
            A = dx.T @ dx / (n - 1)
            w, v = scipy.linalg.eigh(A, B)
            assert np.allclose(mean[k], xk.mean(axis=0), rtol=1e-12)
            assert np.abs(evals[k] / w - 1).max() < 1e-5
            # eigenvectors up to sign, B-normalised: |v_ref' B v| = 1 on the diagonal
            g = np.abs(v.T @ B @ evecs[k])
            assert np.abs(np.diag(g) - 1).max() < 1e-5
        assert np.all(np.diff(evals, axis=1) >= 0)


@pytest.mark.parametrize("field,stride,dt", [(3, 3, np.float64), (2, 1, np.float32), (6, 6, np.uint8), (5, 3, np.float32)])
def test_sfa_train_layer_other_widths(native_lib, field, stride, dt):
    """Nodes narrower than one matrix-core tile (d = 9, 4: padded), overlapping fields, and nodes wider than 16
    inputs (d = 36, 25: the vector kernel); ragged sample counts."""
    from pyfaceanalysis_amd import nodes as N, synth
    from pyfaceanalysis_amd.train import sfa_train_layer
    side = {3: 24, 2: 9, 6: 24, 5: 23}[field]
    n = 1237
    seq = synth.make_training_sequence(n, 32, seed=5)[:, :side * side].copy()
    sb = N.Rectangular2dSwitchboard((side, side), (field, field), (stride, stride), 1)
    d = field * field
    conn = sb.connections.reshape(-1, d)
    evals, evecs, mean, _ = sfa_train_layer(seq.astype(dt), conn)
    assert evals.shape == (conn.shape[0], d)
    for k in sorted({0, conn.shape[0] // 2, conn.shape[0] - 1}):
        xk = seq[:, conn[k]].astype(np.float64)
        B = np.cov(xk.T)
        dx = xk[1:] - xk[:-1]
        A = dx.T @ dx / (n - 1)
        w, v = scipy.linalg.eigh(A, B)
        assert np.allclose(mean[k], xk.mean(axis=0), rtol=1e-12)
        assert np.abs(evals[k] / w - 1).max() < 1e-5
        g = np.abs(v.T @ B @ evecs[k])
        assert np.abs(np.diag(g) - 1).max() < 1e-5
