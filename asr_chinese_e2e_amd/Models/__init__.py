"""Model registry: main.py looks models up by name with getattr (main.py:38-41).  Only the
model that actually runs in the reference is provided plus the CTC-only
variant BASELINE.json asks for."""
from .transformer_official import TransformerCTC, TransformerOffical
