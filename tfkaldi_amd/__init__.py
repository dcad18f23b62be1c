"""tfkaldi_amd: MI355X (gfx950) native engine behind the Nnet / Trainer / Decoder API of vrenkens/tfkaldi.

Layout:
  csrc/            hand-written HIP kernels + the C-ABI engine (include/tfkaldi_hip.h)
  _lib.py          ctypes binding (fails loudly if the library is missing -- no CPU fallback)
  engine.py        numpy-facing handle on one engine
  neuralNetworks/  Nnet, Trainer / CrossEnthropyTrainer, Decoder, classifiers.* (reference API)
  processing/      ark, feature_reader, batchdispenser, target_coder, readfiles (reference I/O)
"""
__version__ = "0.1.0"

import os as _os

# The host driver of the MI355X boxes only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL (and any
# sharing of device memory across processes) fails with `hipIpcGetMemHandle: invalid argument`.  The HSA runtime reads
# the variable when it initialises -- at the first HIP call of the process -- so it is set HERE, at package import,
# before anything of this package touches the GPU (dataparallel.init_from_env checks it again before a multi-rank
# RCCL group is created, for processes that initialised HIP before importing the package).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
