"""nfai_amd — MI355X (gfx950) backend for NFAI's Llama-3 TransformerBlock decode path.

Layout (only what the hot path needs):
  csrc/            hand-written HIP kernels + the C ABI (libnfai_hip.so; header: include/nfai_hip.h)
  _lib.py          ctypes binding (fails loudly when the library is missing)
  hip.py           HipBufferManager / ShaderProperty      ≙ NFAI.Vulkan + ShaderProperty.cs
  shaders.py       the ten op classes + TransformerBlock   ≙ NFAI.Vulkan.Shaders
  llama_model.py   LlamaModel / LlamaModelFactory / SamplingUtils / ModelOptions
  synth.py         synthetic Llama-3 shaped weights for tests and bench (no checkpoints offline)
"""
__version__ = "0.1.0"
