"""Host-side data layer with the reference's interface (processing/ of vrenkens/tfkaldi):
ark I/O, CMVN + splicing feature reader, alignment coder and utterance batch dispenser."""
from . import ark, batchdispenser, feature_reader, readfiles, target_coder  # noqa: F401
