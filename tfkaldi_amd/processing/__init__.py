"""Host-side data layer with the reference's interface (processing/ of vrenkens/tfkaldi):
ark I/O, CMVN + splicing feature reader, alignment coder, utterance batch dispenser, and the feature computation
(sigproc / base / feat / prepare_data) whose arithmetic runs on the GPU."""
from . import ark, batchdispenser, feature_reader, readfiles, target_coder  # noqa: F401
from . import base, feat, prepare_data, sigproc  # noqa: F401
