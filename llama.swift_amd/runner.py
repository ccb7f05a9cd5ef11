"""Python mirror of the reference's public Swift surface (Sources/llama/LlamaRunner.swift:11-124):
``LlamaRunner(modelURL:)``, ``Config(numThreads:numTokens:reversePrompt:)``, ``RunState`` and the
two ``run`` flavours (token stream / token handler), driven through the C mirror of the bridge
(include/llama_runner.h -> csrc/runner.cpp -> include/llamahip.h)."""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import Callable, Iterator, Optional

from . import binding


@dataclass(frozen=True)
class Config:
    """LlamaRunner.Config (LlamaRunner.swift:12-32); default = Config(8, 512, None)."""
    numThreads: int = 8
    numTokens: int = 512
    reversePrompt: Optional[str] = None
    # harness extensions (not part of the Swift surface)
    n_ctx: int = 0
    greedy: bool = False
    seed: int = -1
    keepModel: bool = False      # keep the model loaded between runs of one LlamaRunner


Config.default = Config()  # type: ignore[attr-defined]


class RunState(enum.Enum):
    """LlamaRunner.RunState (LlamaRunner.swift:34-40)."""
    notStarted = 0
    initializing = 1
    generatingOutput = 2
    completed = 3
    failed = 4


class _CConfig(C.Structure):
    _fields_ = [("numberOfThreads", C.c_uint32), ("numberOfTokens", C.c_uint32), ("reversePrompt", C.c_char_p),
                ("n_ctx", C.c_int32), ("greedy", C.c_int32), ("seed", C.c_int32), ("keepModel", C.c_int32)]


_HANDLER = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_int32)


class LlamaRunner:
    def __init__(self, modelURL: str):
        self.modelURL = modelURL
        self._bridge = None          # created with the first run; owns the kept model (Config.keepModel)

    def _get_bridge(self):
        L = binding.lib()
        L.llama_runner_bridge_new.restype = C.c_void_p
        L.llama_runner_bridge_new.argtypes = [C.c_char_p]
        L.llama_runner_bridge_free.argtypes = [C.c_void_p]
        L.llama_runner_bridge_loads.argtypes = [C.c_void_p]
        L.llama_runner_bridge_loads.restype = C.c_int64
        L.llama_runner_bridge_run.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(_CConfig), _HANDLER, C.c_void_p]
        L.llama_runner_bridge_run.restype = C.c_int32
        if self._bridge is None:
            self._bridge = C.c_void_p(L.llama_runner_bridge_new(self.modelURL.encode()))
        return L, self._bridge

    @property
    def loads(self) -> int:
        """Model loads performed so far by this runner's bridge."""
        if self._bridge is None:
            return 0
        return int(binding.lib().llama_runner_bridge_loads(self._bridge))

    def close(self) -> None:
        if self._bridge is not None:
            binding.lib().llama_runner_bridge_free(self._bridge)
            self._bridge = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, prompt: str, config: Config = Config.default,  # type: ignore[attr-defined]
            tokenHandler: Optional[Callable[[bytes], None]] = None,
            stateChangeHandler: Optional[Callable[[RunState, Optional[Exception]], None]] = None) -> list[bytes]:
        """Closure-based run (LlamaRunner.swift:90-123).  Returns the emitted tokens as well."""
        L, bridge = self._get_bridge()

        tokens: list[bytes] = []
        failure: list[Exception] = []

        def notify(state: RunState, error: Optional[Exception] = None) -> None:
            if stateChangeHandler:
                stateChangeHandler(state, error)

        def on_event(_user, etype, text, length, code):
            if etype == 0:
                notify(RunState.initializing)
            elif etype == 2:
                notify(RunState.generatingOutput)
            elif etype == 3:
                tok = C.string_at(text, length) if text else b""
                tokens.append(tok)
                if tokenHandler:
                    tokenHandler(tok)
            elif etype == 4:
                notify(RunState.completed)
            elif etype == 5:
                e = binding.LlamaHipError(code, C.string_at(text, length).decode(errors="replace") if text else "")
                failure.append(e)
                notify(RunState.failed, e)

        notify(RunState.notStarted)                      # LlamaRunner.swift:57 / :96
        cb = _HANDLER(on_event)
        cfg = _CConfig(config.numThreads, config.numTokens,
                       config.reversePrompt.encode() if config.reversePrompt is not None else None,
                       config.n_ctx, int(config.greedy), config.seed, int(config.keepModel))
        L.llama_runner_bridge_run(bridge, prompt.encode(), C.byref(cfg), cb, None)
        if failure:
            raise failure[0]
        return tokens

    def stream(self, prompt: str, config: Config = Config.default,  # type: ignore[attr-defined]
               stateChangeHandler=None) -> Iterator[bytes]:
        """Stream-flavoured run (LlamaRunner.swift:51-87).  The C driver is synchronous, so the
        tokens are produced first and then yielded in order."""
        yield from self.run(prompt, config, None, stateChangeHandler)
