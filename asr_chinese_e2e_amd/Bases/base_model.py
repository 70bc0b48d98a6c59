"""Model plugin surface of the reference (Predictor/Bases/base_model.py:24-72): what main.py and
the trainers rely on - save/load of a plain state_dict, num_para, get_default_config, iterate.
The reference's unused single-process DataParallel `Wrapper` (base_model.py:9-21) is replaced by
one process per GPU over RCCL (asr_chinese_e2e_amd/dist.py)."""
import os

import numpy as np
import torch

from .base_config import BaseConfig


class BaseModel(torch.nn.Module):
    def forward(self, *input):
        raise NotImplementedError

    def iterate(self, *inputs):
        raise NotImplementedError

    def cal_metrics(self, *inputs):
        raise NotImplementedError

    def num_para(self):
        n = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        return f"Trainable parameters:{n}"

    def save(self, path):
        torch.save({k: v.detach().cpu().clone() for k, v in self.state_dict().items()}, path)
        print(f"\nmodel saved to {path}")

    def load(self, path):
        if os.path.isfile(path):
            state = torch.load(path, map_location="cpu", weights_only=True)
            self.load_state_dict(state, strict=False)
            print(f"\nLoaded model state from '{path}'")
        else:
            print(f"\nInvalid model state file: '{path}'")

    @classmethod
    def get_default_config(cls):
        class ModelConfig(BaseConfig):
            pass

        return ModelConfig

    def wrap(self, device_ids=(0, 1)):
        raise RuntimeError("single-process DataParallel is not provided: launch one process per GPU "
                           "(torch.distributed.run) and call asr_chinese_e2e_amd.dist.init()")
