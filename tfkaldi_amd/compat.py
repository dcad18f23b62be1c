"""Make the reference's import lines resolve to this package, so its main.py runs unchanged:

    from neuralNetworks import nnet
    from processing import ark, prepare_data, feature_reader, batchdispenser, target_coder

`install()` registers `neuralNetworks` and `processing` in sys.modules, including the feature computation
(processing/prepare_data.py, feat.py, base.py, sigproc.py: GPU-backed here).  Nothing of the reference checkout is
ever put on a package path: a module this package does not provide fails to import.
"""
import importlib
import sys


def install():
    from . import neuralNetworks, processing
    sys.modules["neuralNetworks"] = neuralNetworks
    sys.modules["processing"] = processing
    for pkg, names in (("neuralNetworks", ("nnet", "trainer", "decoder", "classifiers")),
                       ("processing", ("ark", "feature_reader", "batchdispenser", "target_coder", "readfiles", "prepare_data",
                                      "feat", "base", "sigproc"))):
        for name in names:
            sys.modules["%s.%s" % (pkg, name)] = importlib.import_module("tfkaldi_amd.%s.%s" % (pkg, name))
