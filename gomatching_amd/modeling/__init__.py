from .meta_arch import GoMatching  # noqa: F401
from .roi_heads import LSTMatcher, SHA_FFN_CRSATTN, build_roi_heads  # noqa: F401
from .deepsolo import DeepSolo  # noqa: F401
from .backbone import ResNet50  # noqa: F401
