"""CPU oracle for the argsim sequence-VAE ELBO step -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the reported CPU baseline.

PARITY STATUS: **parity unpinned** for the model graph.  The reference
(`/root/reference/src/model.py`) needs TensorFlow 1.x + ``tf.contrib.cudnn_rnn``
which is not installed (plain ModuleNotFoundError, nothing was denied) and it
ships no golden vectors, tests, fixtures or checkpoints for this path
(SURVEY.md section 8c).  The oracle is therefore a line-by-line restatement of
``model.py:75-189`` plus the published cuDNN GRU equations, pinned only by
  * the schedule table in ``docs/log.org:21-28`` (keepwd / anneal columns),
  * the embedding-bound identity in ``docs/log.org:86-88``,
  * agreement between two independent restatements (numpy fp64 loops here and
    a torch autograd twin built on ``torch.nn.functional`` primitives),
  * ``torch.nn.GRU`` (CPU) which implements the same reset-after equations.
The pure helpers ``vpack / partition / sample`` (``util_np.py:5-33``) ARE pinned:
golden vectors were captured by importing the reference module in the build
container (``tests/golden/make_util_np_golden.py``).
"""
