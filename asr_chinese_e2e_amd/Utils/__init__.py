from .pack import Pack
from .score import calculate_cer, edit_distance
from .tfevents import EventFileWriter, read_events
