from .vocab import Vocab
from .padder import Padder
from .batch import collat, synthetic_pack, DataConfigAiShell1
from .processor import AudioParser, build_LFR_features
from .loader import BucketedWaveLoader, WaveDataset, bucket_batches, build_dataloader, load_wav, shard_batches
