from .trainer import HipSGD, build_optimizer, run_step, wrap_model_with_ddp  # noqa: F401
