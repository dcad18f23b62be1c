"""Make the reference's import lines resolve to this package, so its main.py runs unchanged:

    from neuralNetworks import nnet
    from processing import ark, prepare_data, feature_reader, batchdispenser, target_coder

`install()` registers `neuralNetworks` and `processing` in sys.modules, including the feature computation
(processing/prepare_data.py, feat.py, base.py, sigproc.py: GPU-backed here).  Modules this package does not
provide (`target_normalizers` variants of other corpora, ...) are still found in the reference checkout when
`reference_root` is given.
"""
import importlib
import os
import sys


def install(reference_root=None):
    from . import neuralNetworks, processing
    sys.modules["neuralNetworks"] = neuralNetworks
    sys.modules["processing"] = processing
    for pkg, names in (("neuralNetworks", ("nnet", "trainer", "decoder", "classifiers")),
                       ("processing", ("ark", "feature_reader", "batchdispenser", "target_coder", "readfiles", "prepare_data",
                                      "feat", "base", "sigproc"))):
        for name in names:
            sys.modules["%s.%s" % (pkg, name)] = importlib.import_module("tfkaldi_amd.%s.%s" % (pkg, name))
    if reference_root:
        ref_processing = os.path.join(reference_root, "processing")
        if os.path.isdir(ref_processing) and ref_processing not in processing.__path__:
            processing.__path__.append(ref_processing)
