from .synthetic import make_batch, make_class_embeddings  # noqa: F401
