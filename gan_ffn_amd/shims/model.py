"""`model` shim: put this directory FIRST on sys.path (or `sys.modules["model"] = <this module>`) and the reference's
trainers bind the MI355X implementation without a source change:

    /root/reference/train_IEMOCAP.py:18-29              from model import MaskedNLLLoss, FocalLoss, LSTMModel2,
                                                             AcousticGenerator, ..., VisualDiscriminator, GAN_FFN
    /root/reference/train_IEMOCAP_DialogueRNN.py:18-30  ... + GAN_FFN_DialogueRNN

The hot-path classes come from gan_ffn_amd.model (same names, constructors, forward signatures, state_dict keys; all
arithmetic in libganffn.so).  `FocalLoss` and `LSTMModel2` are imported by the script but never used (their only call
sites are commented out, train_IEMOCAP.py:640-643,657): if the maintainer keeps the reference's own file importable as
`model_reference`, they are taken from there, otherwise they are placeholders that fail loudly on construction.
"""
from gan_ffn_amd.model import (PositionalEncoding, AcousticGenerator, VisualGenerator, TextGenerator,       # noqa: F401
                               AcousticDiscriminator, VisualDiscriminator, TextDiscriminator, GAN_FFN, MaskedNLLLoss)
from gan_ffn_amd.model import (GAN_FFN_DialogueRNN, BiModel, DialogueRNN, DialogueRNNCell, MatchingAttention,   # noqa: F401
                               SimpleAttention, MELDLSTMModel)

try:                                                    # the reference's legacy classes, if its file is kept around
    from model_reference import FocalLoss, LSTMModel2   # noqa: F401
except ImportError:
    class _OffPath:
        """stands in for a reference class that is off the generator/discriminator path"""

        def __init__(self, *a, **kw):
            raise NotImplementedError("%s is not part of the GAN-FFN hot path; keep the reference's model.py importable as "
                                      "`model_reference` to use it" % type(self).__name__)

    class FocalLoss(_OffPath):
        pass

    class LSTMModel2(_OffPath):
        pass
