"""automerge_amd — MI355X-native batched load engine for Automerge documents.

Only what the hot path needs: csrc/ (HIP kernels + C ABI, libautomerge_hip.so) and this host-side
mirror of the reference's bulk `Automerge` API.  Importing `automerge_amd.api.Engine` requires the
HIP library; there is no CPU fallback."""
from .api import AutomergeError, Batch, Engine, EngineError, SyncHave, SyncMessage, SyncState, load_library  # noqa: F401
