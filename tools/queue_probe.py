"""What does a torch-captured hipGraph make of the tile operator's work queue?  ("the replay abort".)

Two inspections, neither of which runs a tile kernel on a doubtful queue:

1. `stair_debug_queue_probe` under torch.cuda.graph: the reset + the ticket protocol of stair_plan_run with a kernel that only
   RECORDS the tickets it draws.  Three layouts: round 3's (hipMemsetAsync reset, one head per launch, no self-reset), today's
   (zeroing kernel, one shared self-resetting pair) and the self-resetting pair with no reset at all.  Each graph is replayed
   four times with unrelated eager work in between; after every replay the workgroups of every launch must have taken exactly
   `total` tiles between them, with no ticket beyond total + grid.
   Plus: a graph holding ONE captured reset (hipMemsetAsync or the zero-fill kernel) of 256 B ... 1 MiB, target refilled with
   ones before every replay.
2. The node list of those graphs and of a real captured plan (hipGraphGetNodes / NodeGetType / MemsetNodeGetParams /
   NodeGetDependencies through ctypes on libamdhip64): is the reset a memset node, what are its dst / width / element size, does
   every kernel node sit behind it in one chain?

Writes one JSON document (argv[1], default gpurun_out/r04_queue_probe.json)."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stair_amd import spec, synth                                  # noqa: E402
from stair_amd._lib import lib, check                              # noqa: E402
from stair_amd.module_net import VideoNMN                          # noqa: E402

hip = C.CDLL('libamdhip64.so')


class MemsetParams(C.Structure):
    _fields_ = [('dst', C.c_void_p), ('elementSize', C.c_uint), ('height', C.c_size_t), ('pitch', C.c_size_t),
                ('value', C.c_uint), ('width', C.c_size_t)]


class Dim3(C.Structure):
    _fields_ = [('x', C.c_uint), ('y', C.c_uint), ('z', C.c_uint)]


class KernelParams(C.Structure):
    _fields_ = [('blockDim', Dim3), ('extra', C.c_void_p), ('func', C.c_void_p), ('gridDim', Dim3),
                ('kernelParams', C.c_void_p), ('sharedMemBytes', C.c_uint)]


NODE_TYPES = {0: 'kernel', 1: 'memcpy', 2: 'memset', 3: 'host', 4: 'graph', 5: 'empty', 6: 'wait_event', 7: 'event_record'}


def graph_nodes(graph):
    """[(type, params, [indices of dependencies])] of a torch.cuda.CUDAGraph created with keep_graph=True."""
    g = C.c_void_p(int(graph.raw_cuda_graph()))
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(g, None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(g, nodes, C.byref(n)) == 0
    index = {nodes[i]: i for i in range(n.value)}
    out = []
    for i in range(n.value):
        node = C.c_void_p(nodes[i])
        t = C.c_int(-1)
        assert hip.hipGraphNodeGetType(node, C.byref(t)) == 0
        info = {'type': NODE_TYPES.get(t.value, str(t.value))}
        if t.value == 2:
            mp = MemsetParams()
            assert hip.hipGraphMemsetNodeGetParams(node, C.byref(mp)) == 0
            info.update(dst=mp.dst, element_size=mp.elementSize, width=mp.width, height=mp.height, value=mp.value)
        elif t.value == 0:
            kp = KernelParams()
            if hip.hipGraphKernelNodeGetParams(node, C.byref(kp)) == 0:
                info.update(grid=[kp.gridDim.x, kp.gridDim.y, kp.gridDim.z], block=kp.blockDim.x, lds=kp.sharedMemBytes)
        nd = C.c_size_t(0)
        assert hip.hipGraphNodeGetDependencies(node, None, C.byref(nd)) == 0
        deps = (C.c_void_p * max(nd.value, 1))()
        if nd.value:
            assert hip.hipGraphNodeGetDependencies(node, deps, C.byref(nd)) == 0
        info['deps'] = [index[deps[j]] for j in range(nd.value)]
        out.append(info)
    return out


def chain_summary(nodes):
    """Is the graph one chain?  Returns the counts per type and, for every memset node, how many kernel nodes are (transitively)
    behind it."""
    n = len(nodes)
    children = [[] for _ in range(n)]
    for i, nd in enumerate(nodes):
        for d in nd['deps']:
            children[d].append(i)
    types = {}
    for nd in nodes:
        types[nd['type']] = types.get(nd['type'], 0) + 1
    roots = [i for i, nd in enumerate(nodes) if not nd['deps']]
    linear = len(roots) == 1 and all(len(c) <= 1 for c in children) and all(len(nd['deps']) <= 1 for nd in nodes)
    memsets = []
    for i, nd in enumerate(nodes):
        if nd['type'] != 'memset':
            continue
        seen, stack = set(), [i]
        while stack:
            for c in children[stack.pop()]:
                if c not in seen:
                    seen.add(c)
                    stack.append(c)
        memsets.append({'node': i, 'dst': nd['dst'], 'width': nd['width'], 'height': nd['height'], 'element_size': nd['element_size'],
                        'value': nd['value'], 'kernels_behind': sum(nodes[j]['type'] == 'kernel' for j in seen),
                        'kernels_total': types.get('kernel', 0), 'deps': nd['deps']})
    return {'nodes': n, 'types': types, 'roots': len(roots), 'linear_chain': linear, 'memset_nodes': memsets}


def probe(name, reset_mode, per_launch_heads, self_reset, launches=6, grid=256, total=273, replays=4):
    dev = torch.device('cuda:0')
    words = torch.full((128,), 7, dtype=torch.int32, device=dev)         # "a workspace that has been used for something else"
    seen = torch.zeros(launches, grid, 2, dtype=torch.int32, device=dev)
    noise = torch.randn(1024, 1024, device=dev)

    def enqueue(stream):
        check(lib.stair_debug_queue_probe(C.c_void_p(words.data_ptr()), C.c_void_p(seen.data_ptr()), launches, grid, total, reset_mode,
                                          per_launch_heads, self_reset, C.c_void_p(stream.cuda_stream)))

    if reset_mode == 2:
        words.zero_()                         # the self-resetting pair needs ONE zeroing in its life
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        enqueue(side)                         # eager warm-up, as CapturedPlan does
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(graph):
        enqueue(torch.cuda.current_stream(dev))
    summary = chain_summary(graph_nodes(graph))
    words_addr = words.data_ptr()
    for m in summary['memset_nodes']:
        m['dst_is_queue_words'] = m['dst'] == words_addr
    graph.instantiate()
    rows = []
    for r in range(replays):
        seen.zero_()
        (noise @ noise).sum().item()          # unrelated eager work between two replays, as a serving loop has
        graph.replay()
        torch.cuda.synchronize(dev)
        s = seen.cpu().numpy().astype('int64') & 0xffffffff
        # a launch is right when its workgroups took exactly the `total` tiles between them and nobody drew a ticket beyond
        # total + grid (first tickets need not be a permutation: a fast workgroup draws several before a slow one draws its first)
        ok = [int(s[l, :, 1].sum()) == total and int(s[l, :, 0].max()) < total + grid for l in range(launches)]
        w = (words.cpu().numpy().astype('int64') & 0xffffffff)[16:16 + 2 * launches].tolist()
        rows.append({'replay': r, 'launches_ok': ok, 'min_first_ticket': int(s[:, :, 0].min()), 'max_first_ticket': int(s[:, :, 0].max()),
                     'tiles_taken_per_launch': s[:, :, 1].sum(1).tolist(), 'queue_words_after': w})
    return {'layout': name, 'reset_mode': ['hipMemsetAsync', 'kernel', 'none'][reset_mode], 'per_launch_heads': bool(per_launch_heads),
            'self_reset': bool(self_reset), 'launches': launches, 'grid': grid, 'total': total, 'graph': summary, 'replays': rows,
            'all_ok': all(all(r['launches_ok']) for r in rows)}


def memset_sweep(mode, sizes=(256, 768, 4096, 49152, 1 << 20), replays=4):
    """A captured graph holding ONE reset of `bytes` bytes (mode 0: hipMemsetAsync, 1: the zero-fill kernel): the buffer is filled with
    ones before every replay, unrelated eager launches run in between; what does the buffer hold afterwards?"""
    dev = torch.device('cuda:0')
    rows = []
    noise = torch.randn(512, 512, device=dev)
    for nbytes in sizes:
        buf = torch.ones(nbytes // 4, dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            check(lib.stair_debug_memset(C.c_void_p(buf.data_ptr()), nbytes, mode, C.c_void_p(side.cuda_stream)))
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(graph):
            check(lib.stair_debug_memset(C.c_void_p(buf.data_ptr()), nbytes, mode, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        types = chain_summary(graph_nodes(graph))['types']
        graph.instantiate()
        per_replay = []
        for r in range(replays):
            buf.fill_(1)
            (noise @ noise).sum().item()
            torch.zeros(1000, device=dev).add_(1.0)
            graph.replay()
            torch.cuda.synchronize(dev)
            b = buf.cpu().numpy().astype('int64') & 0xffffffff
            nz = int((b != 0).sum())
            per_replay.append({'replay': r, 'nonzero_words': nz, 'first_words_hex': [hex(int(v)) for v in b[:4]] if nz else []})
        rows.append({'bytes': nbytes, 'graph_nodes': types, 'replays': per_replay, 'all_zero_every_replay': all(x['nonzero_words'] == 0 for x in per_replay)})
    return rows


def captured_plan():
    config = dict(spec.DEFAULT_CONFIG)
    torch.manual_seed(4)
    model = VideoNMN(config).to('cuda:0')
    qs = synth.make_questions(config, 31, 24, forms=synth.ALL_FORMS)
    res = model.forward_batch(qs)
    from stair_amd import module_net
    cap = module_net.CapturedPlan.__new__(module_net.CapturedPlan)
    # CapturedPlan.__init__ with keep_graph=True so that the recorded graph can be listed
    cap.result = res
    info, dev = res.info, res._video.device
    cap._ws = torch.empty((info.workspace_bytes + 3) // 4, dtype=torch.float32, device=dev)
    cap.logits, cap.pred = res.logits, res.pred
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        check(lib.stair_plan_upload(res._plan, C.c_void_p(cap._ws.data_ptr()), cap._ws.numel() * 4, C.c_void_p(side.cuda_stream)))
        cap._enqueue(side)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    cap.graph = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(cap.graph):
        cap._enqueue(torch.cuda.current_stream(dev))
    nodes = graph_nodes(cap.graph)
    summary = chain_summary(nodes)
    ws0, ws1 = cap._ws.data_ptr(), cap._ws.data_ptr() + cap._ws.numel() * 4
    status = ws0 + 4 * info.status_off
    for m in summary['memset_nodes']:
        m['inside_workspace'] = ws0 <= m['dst'] < ws1
        m['is_status_block'] = m['dst'] == status
    summary['tile_kernel_nodes'] = sum(1 for nd in nodes if nd['type'] == 'kernel' and nd.get('block') == 512 and nd.get('lds', 0) > 100000)
    return summary


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r04_queue_probe.json'
    doc = {'torch': torch.__version__, 'hip': torch.version.hip, 'device': torch.cuda.get_device_name(0),
           'probes': [probe('round 3: memset reset, one head per launch', 0, 1, 0),
                      probe('round 4: zeroing kernel, one shared self-resetting pair', 1, 0, 1),
                      probe('self-resetting pair, no reset in the graph', 2, 0, 1)],
           'captured_memset_by_size': {'hipMemsetAsync': memset_sweep(0), 'zero_fill_kernel': memset_sweep(1)},
           'captured_plan_graph': captured_plan()}
    os.makedirs(os.path.dirname(out) or '.', exist_ok=True)
    with open(out, 'w') as f:
        json.dump(doc, f, indent=1)
    for p in doc['probes']:
        print(p['layout'], '->', 'ok' if p['all_ok'] else 'TICKETS WRONG', '| graph:', p['graph']['types'], 'linear' if p['graph']['linear_chain'] else 'NOT linear')
        for m in p['graph']['memset_nodes']:
            print('   memset node', m)
    for k, rows in doc['captured_memset_by_size'].items():
        for r in rows:
            print(k, r['bytes'], 'bytes:', r['graph_nodes'], 'zero after every replay' if r['all_zero_every_replay'] else
                  'NOT ZERO: %s' % [(x['replay'], x['nonzero_words'], x['first_words_hex']) for x in r['replays'] if x['nonzero_words']])
    print('captured plan:', doc['captured_plan_graph']['types'], 'linear' if doc['captured_plan_graph']['linear_chain'] else 'NOT linear',
          'tile kernels', doc['captured_plan_graph']['tile_kernel_nodes'])
    for m in doc['captured_plan_graph']['memset_nodes']:
        print('   memset node', m)


if __name__ == '__main__':
    main()
