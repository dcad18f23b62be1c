"""The CTC oracle (oracle/ctc_oracle.py) pinned against an independent implementation: torch's CPU ctc_loss with
the blank moved to the last class, as TensorFlow's op has it.  Loss and gradient w.r.t. the logits."""
import numpy as np
import pytest

from oracle.ctc_oracle import ctc_batch, ctc_loss_and_grad


def _torch_ctc(logits, labels):
    import torch
    z = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    lp = torch.log_softmax(z, dim=-1).unsqueeze(1)  # [T, 1, O]
    loss = torch.nn.functional.ctc_loss(lp, torch.tensor([labels], dtype=torch.long), torch.tensor([z.shape[0]]),
                                        torch.tensor([len(labels)]), blank=z.shape[1] - 1, reduction="sum",
                                        zero_infinity=False)
    loss.backward()
    return float(loss.detach()), z.grad.numpy()


@pytest.mark.parametrize("T,O,labels", [
    (12, 6, [0, 1, 2]),
    (9, 5, [1, 1, 2, 2]),        # repeated labels need a blank between them
    (7, 4, [0, 0, 0]),           # exactly feasible: T = S + repeats
    (15, 8, []),                 # empty target: all blanks
    (20, 30, [3, 7, 7, 2, 28, 0, 3]),
    (1, 3, [1]),
])
def test_ctc_oracle_matches_torch(T, O, labels):
    rng = np.random.default_rng(T * 100 + O)
    logits = rng.standard_normal((T, O)) * 2
    loss, grad = ctc_loss_and_grad(logits, labels)
    want_loss, want_grad = _torch_ctc(logits, labels)
    assert np.isfinite(loss)
    np.testing.assert_allclose(loss, want_loss, rtol=1e-10)
    np.testing.assert_allclose(grad, want_grad, rtol=1e-8, atol=1e-12)
    # rows of the gradient sum to zero (softmax minus a distribution over the classes)
    assert np.abs(grad.sum(axis=1)).max() < 1e-10


def test_ctc_oracle_infeasible_and_batch():
    rng = np.random.default_rng(3)
    loss, grad = ctc_loss_and_grad(rng.standard_normal((3, 5)), [1, 1, 2])  # needs 4 frames
    assert loss == np.inf and not grad.any()
    logits = rng.standard_normal((11 + 6, 7))
    total, grad, n_labels = ctc_batch(logits, [11, 6], [0, 1, 2, 5, 5], [3, 2])
    l0, g0 = ctc_loss_and_grad(logits[:11], [0, 1, 2])
    l1, g1 = ctc_loss_and_grad(logits[11:], [5, 5])
    assert n_labels == 5 and np.isclose(total, l0 + l1)
    assert (grad[:11] == g0).all() and (grad[11:] == g1).all()
