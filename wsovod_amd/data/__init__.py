from .synthetic import make_batch, make_class_embeddings  # noqa: F401
from .multi_dataset import (MultiDatasetAspectRatioGroupedDataset, MultiDatasetTrainingSampler,  # noqa: F401
                            repeat_factors_from_category_frequency)
from .proposals import (load_class_embeddings, load_d2_pickle_into, load_proposals_into_dataset,  # noqa: F401
                        transform_proposals, unique_boxes)
