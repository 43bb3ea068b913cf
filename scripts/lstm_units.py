"""pa_lstm_forward alone (64 windows x 7 frames, H = 512, 3 layers) for each PA_LSTM_UNITS setting: us per call and per step."""
import ctypes as C, os, subprocess, sys
if len(sys.argv) > 1:
    os.environ["PA_LSTM_UNITS"] = sys.argv[1]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, time
    from playaid_core_amd import synth, _lib
    from playaid_core_amd.anim_ontology import ACTIONS
    from playaid_core_amd.rnn_action_detector import RNNActionDetector
    m = RNNActionDetector("Joker", ACTIONS, state_dict=synth.make_rnn_state_dict(), max_rows=64 * 7)
    feats = torch.randn((64 * 7, _lib.PA_FEATURE_STRIDE), device="cuda")
    out = torch.empty((64 * 7, m.num_actions), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run(L):
        rc = m._lib.pa_lstm_forward(m._h, C.c_void_p(feats.data_ptr()), _lib.PA_FEATURE_STRIDE, L, 7, C.c_void_p(out.data_ptr()), st)
        assert rc == 0
    def timed(L):
        for _ in range(3): run(L)
        torch.cuda.synchronize(); m.check()
        t0 = time.perf_counter()
        for _ in range(20): run(L)
        torch.cuda.synchronize(); m.check()
        return (time.perf_counter() - t0) / 20
    d64, d16 = timed(64), timed(16)
    step = (d64 - d16) / (48 * 3)
    print(f"units {sys.argv[1]}: {d64 * 1e6:8.1f} us per 64-step call, {d16 * 1e6:8.1f} per 16-step call -> {step * 1e6:6.2f} us per (layer, step), "
          f"{(d64 - step * 192) * 1e6:7.1f} us fixed per call; finite {bool(torch.isfinite(out).all())}")
else:
    for u in ("8", "4", "2", "1"):
        subprocess.run([sys.executable, __file__, u])
