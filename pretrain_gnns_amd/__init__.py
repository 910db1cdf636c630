"""MI355X-native hot path of snap-stanford/pretrain-gnns (GIN/GCN message passing)."""
__version__ = "0.1.0"
