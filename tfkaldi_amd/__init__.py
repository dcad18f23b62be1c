"""tfkaldi_amd: MI355X (gfx950) native engine behind the Nnet / Trainer / Decoder API of vrenkens/tfkaldi.

Layout:
  csrc/            hand-written HIP kernels + the C-ABI engine (include/tfkaldi_hip.h)
  _lib.py          ctypes binding (fails loudly if the library is missing -- no CPU fallback)
  engine.py        numpy-facing handle on one engine
  neuralNetworks/  Nnet, Trainer / CrossEnthropyTrainer, Decoder, classifiers.* (reference API)
  processing/      ark, feature_reader, batchdispenser, target_coder, readfiles (reference I/O)
"""
__version__ = "0.1.0"
