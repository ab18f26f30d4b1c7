from .trainer import HipSGD, HotPathTrainer, build_optimizer, run_step, wrap_model_with_ddp  # noqa: F401
