"""utils/hipgraph.GraphEvent: event-record nodes inside a captured hipGraph that code outside the graph can wait for and
time (the front-end pipeline and the exchange schedule of the replayed step are built on it)."""
import pytest


@pytest.mark.gpu
def test_event_node_orders_another_stream_and_times_the_graph(hip):
    import torch
    from mxdetection_amd.utils.hipgraph import GraphEvent
    n = 1 << 24
    a = torch.zeros((n,), device="cuda")
    b = torch.zeros((n,), device="cuda")
    out = torch.zeros((n,), device="cuda")
    side = torch.cuda.Stream()
    e0, e1 = GraphEvent(), GraphEvent()
    with pytest.raises(RuntimeError, match="not capturing"):
        e0.record_node()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        g.capture_begin(capture_error_mode="thread_local")
        e0.record_node()
        for _ in range(8):
            a.add_(1.0)                 # the work the side stream must see completed
        e1.record_node()
        for _ in range(8):
            b.add_(1.0)                 # more work behind the mark: the side stream does not wait for it
        g.capture_end()
    torch.cuda.current_stream().wait_stream(cap)
    for rep in range(1, 4):
        g.replay()
        e1.wait(side)                   # the record of THIS launch
        with torch.cuda.stream(side):
            out.copy_(a)
        side.synchronize()
        assert float(out[0]) == 8.0 * rep and float(out[-1]) == 8.0 * rep, rep
        torch.cuda.synchronize()
    e1.synchronize()
    us = e0.elapsed_us(e1)              # device time of the eight kernels between the two nodes (64 MiB read + written each)
    assert 20.0 < us < 20000.0, us
    # plain records on a stream that is not capturing work too
    p0, p1 = GraphEvent(), GraphEvent()
    p0.record()
    a.add_(1.0)
    p1.record()
    p1.synchronize()
    assert p0.elapsed_us(p1) > 0.0
