from .optimizer import FusedAdam, NoamOpt
from .trainer11 import Trainer11
from .base_trainer import BaseTrainer
