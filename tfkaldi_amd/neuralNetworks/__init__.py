"""Nnet / Trainer / Decoder / classifiers with the reference's Python API (neuralNetworks/ of
vrenkens/tfkaldi); the TensorFlow graph + session is replaced by the C-ABI HIP engine."""
