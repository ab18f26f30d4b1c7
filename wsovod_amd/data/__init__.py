from .synthetic import make_batch, make_class_embeddings  # noqa: F401
from .multi_dataset import (MultiDatasetAspectRatioGroupedDataset, MultiDatasetTrainingSampler,  # noqa: F401
                            repeat_factors_from_category_frequency)
