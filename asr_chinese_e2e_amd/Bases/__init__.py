from .base_config import BaseConfig
from .base_model import BaseModel
