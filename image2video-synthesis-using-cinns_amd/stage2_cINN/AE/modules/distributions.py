"""The part of the reference's ``stage2_cINN/AE/modules/distributions.py`` that the sampling path touches:
``ResnetEncoder.encode(x).mode()`` (INN.py:62).  ``mode()`` of the diagonal Gaussian is its mean = the first E of the 2E
encoder channels (distributions.py:9,41-42); the native embedder only evaluates those E channels."""


class PosteriorMean(object):
    def __init__(self, mean):
        self.mean = mean            # [B, E, 1, 1]
        self.deterministic = True

    def mode(self):
        return self.mean

    def sample(self):
        raise NotImplementedError("the sampling path only uses .mode(); the log-variance half of the encoder is not evaluated")
