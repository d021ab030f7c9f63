"""Loader of the in-tree HIP library. There is NO fallback: if csrc/libevomg.so is missing or does
not load, importing code gets a RuntimeError telling how to build it."""
import os

from evostencils_amd import abi

# EVOMG_LIB: another build of the same library (A/B experiments of tools/, e.g. a -D variant); the default is the in-tree one
LIB_PATH = os.environ.get("EVOMG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libevomg.so")
_library = None


def library() -> abi.Library:
    global _library
    if _library is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"HIP backend not built: {LIB_PATH} is missing. "
                               f"Run `python -m evostencils_amd.build` (needs hipcc).")
        try:
            _library = abi.Library(LIB_PATH)
        except OSError as exc:
            raise RuntimeError(f"HIP backend failed to load ({LIB_PATH}): {exc}") from exc
        except AttributeError as exc:     # a library built before a symbol of the current header existed
            raise RuntimeError(f"HIP backend {LIB_PATH} lacks a symbol of include/evomg.h ({exc}); "
                               f"rebuild with `python -m evostencils_amd.build`") from exc
        if _library.fn["abi_version"]() != abi.ABI_VERSION:
            raise RuntimeError("HIP backend ABI version mismatch; rebuild with `python -m evostencils_amd.build`")
    return _library


def create(problem_description, device_id=0, parameters=None, jacobi_semantics=None) -> abi.Handle:
    """Native solver instance with the problem's initial fields uploaded. `jacobi_semantics`: None = the
    descriptor's, else "textbook" | "exastencils-v1.1" (abi.JACOBI_SEMANTICS)."""
    h = abi.Handle(library(), problem_description.to_c(parameters, jacobi_semantics), device_id)
    problem_description.upload_initial_fields(h)
    return h


def communicator(rank: int, world: int, device_id: int, id_path: str, timeout_ms: int = 60000) -> abi.Communicator:
    """RCCL communicator behind the C-ABI (one per process / GPU) for the per-generation fitness exchange."""
    return abi.Communicator(library(), rank, world, device_id, id_path, timeout_ms)
