"""Classifier descriptions (reference neuralNetworks/classifiers/): the objects carry the network
structure; the arithmetic runs in the HIP engine."""
