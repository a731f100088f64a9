"""Checkpoint reader -- role of reference utils/common/setup_helper.py:25-30."""
import torch


def load_weights(weights_dir, device):
    """torch.load of a reference checkpoint.  Checkpoints pickle an argparse.Namespace
    ('regressor_config', utils/train/helper.py:15), hence weights_only=False."""
    if weights_dir is None:
        return None
    return torch.load(weights_dir, map_location="cpu", weights_only=False)
