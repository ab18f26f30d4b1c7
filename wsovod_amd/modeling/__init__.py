from .backbone import build_wsl_resnet_backbone  # noqa: F401  (registers)
from .box_head import DiscriminativeAdaptationNeck  # noqa: F401
from .roi_heads import WSOVODMixedDatasetsROIHeads, WSOVODROIHeads  # noqa: F401
from .meta_arch import GeneralizedRCNN_WSOVOD, GeneralizedRCNN_WSOVOD_MixedDatasets, build_model  # noqa: F401
from .proposal_generator import StandardRPNHead, WSOVODRPN_V2, build_proposal_generator  # noqa: F401
from .anchor_generator import DefaultAnchorGenerator  # noqa: F401
from .test_time_augmentation import GeneralizedRCNNWithTTAAVG, GeneralizedRCNNWithTTAUNION  # noqa: F401
