"""Evaluator-side batching.  The reference's evaluators call the network one image at a time
(utils/vocapi_evaluator_mask.py:57-82, utils/vocapi_evaluator.py, utils/cocoapi_evaluator.py:70-98) and rescale the
boxes on the host.  These helpers produce the same data structures with batched forwards and the rescale on the GPU
(`forward_batch(..., sizes_wh=...)` -> y355_scale_boxes); the mAP computation that follows them is untouched.

    # utils/vocapi_evaluator_mask.py:49-95  (evaluate)
    - for i in range(num_images): ... bboxes, scores, cls_inds = net(x, quantization=..., find=...) ...
    + self.all_boxes = voc_all_boxes(net, self.dataset, len(self.labelmap), batch_size=64, quantization=quantization, find=find)

    # utils/cocoapi_evaluator.py:66-98
    + ids, data_dict = coco_data_dict(model, self.dataset, self.transform, batch_size=64)

Both helpers are software-pipelined over the y355_pipeline behind a calibrated q_bf model (`_submit`): batch k + 1 is submitted
before batch k is unpacked, so the GPU runs while the host loads images and builds the evaluator's lists.
"""
import numpy as np
import torch

from .. import _ffi


def _batches(n, bs):
    for i0 in range(0, n, bs):
        yield i0, min(n, i0 + bs)


def _run(net, x, sizes_wh, **kw):
    """One batched forward with the evaluators' rescale.  Only the keyword arguments the model's forward_batch declares
    are passed (decided from its signature, never by catching TypeError: an exception raised inside a forward must
    surface); asking a model for `quantization=True` / `find=True` that it cannot honour is an error, not a silent
    fp32 evaluation.  A q_bf model whose trackers are still un-calibrated is first run on image 0 alone: the
    reference's per-image loop freezes every tracker on its first image (models/slim_yolo_v2.py:25-27), a batched
    first call would calibrate on the maximum over the whole batch."""
    import inspect
    if not hasattr(net, "forward_batch"):
        raise TypeError("evaluator batching needs a yolo355 model (forward_batch)")
    params = inspect.signature(net.forward_batch).parameters
    for k in ("quantization", "find"):
        if kw.get(k) and k not in params:
            raise TypeError("%s.forward_batch does not take %s=True" % (type(net).__name__, k))
    call = {k: v for k, v in kw.items() if k in params}
    if call.get("quantization") and hasattr(net, "_tracker_states") and any(t.first_a == 0 for t in net._tracker_states()):
        net.forward_batch(x[:1], **call)
    if "sizes_wh" in params:
        return net.forward_batch(x, sizes_wh=sizes_wh, **call)
    # models without the fused rescale (the composed wider families): rescale like the reference, per image
    res = []
    for (b, s, c), (w, h) in zip(net.forward_batch(x, **call), sizes_wh):
        b = b.copy()
        b *= np.array([[w, h, w, h]])
        res.append((b, s, c))
    return res


def _submit(net, x, sizes_wh, **kw):
    """Start one batch and return a zero-argument function that delivers its detections.  A calibrated q_bf model
    (quantization=True, no guard) goes through submit_batch / collect_batch -- the y355_pipeline behind the model: the GPU works
    on this batch while the caller loads the next one and unpacks the previous one; everything else runs synchronously in _run."""
    if (kw.get("quantization") and not kw.get("find") and hasattr(net, "submit_batch") and hasattr(net, "_tracker_states")
            and all(t.first_a != 0 for t in net._tracker_states()) and int(x.shape[0]) <= 2 * _ffi.PIPE_DEFAULT_HANDLES * getattr(net, "PIPELINE_CHUNK", 0)):
        token = net.submit_batch(x, quantization=True, find=False, sizes_wh=sizes_wh)
        return lambda: net.collect_batch(token)
    dets = _run(net, x, sizes_wh, **kw)
    return lambda: dets


def voc_all_boxes(net, dataset, num_classes, batch_size=64, quantization=False, find=False, num_images=None):
    """all_boxes[cls][image] = N x 5 float32 (x1, y1, x2, y2, score) exactly as the loop of
    utils/vocapi_evaluator_mask.py:57-82 builds it; dataset.pull_item(i) -> (im [3,H,W] tensor, gt, h, w)."""
    n = len(dataset) if num_images is None else int(num_images)
    all_boxes = [[[] for _ in range(n)] for _ in range(num_classes)]
    def unpack(i0, dets):
        for k, (bboxes, scores, cls_inds) in enumerate(dets):
            i = i0 + k
            for j in range(num_classes):
                inds = np.where(cls_inds == j)[0]
                if len(inds) == 0:
                    all_boxes[j][i] = np.empty([0, 5], dtype=np.float32)
                    continue
                all_boxes[j][i] = np.hstack((bboxes[inds], scores[inds][:, np.newaxis])).astype(np.float32, copy=False)
    pending = None                                      # (first image, deliver) of the batch the GPU is working on
    for i0, i1 in _batches(n, batch_size):
        ims, sizes = [], []
        for i in range(i0, i1):
            im, gt, h, w = dataset.pull_item(i)
            ims.append(torch.as_tensor(im))
            sizes.append((w, h))
        x = torch.stack(ims).float()
        deliver = _submit(net, x, np.asarray(sizes, np.float32), quantization=quantization, find=find)
        if pending is not None:
            unpack(pending[0], pending[1]())
        pending = (i0, deliver)
    if pending is not None:
        unpack(pending[0], pending[1]())
    return all_boxes


def coco_data_dict(net, dataset, transform, batch_size=64, num_images=None, **kw):
    """(ids, data_dict) exactly as utils/cocoapi_evaluator.py:66-98 builds them: dataset.pull_image(i) -> (img HWC BGR,
    id); transform(img)[0] -> HWC float image at the network size; dataset.class_ids maps class index -> COCO id."""
    n = len(dataset) if num_images is None else int(num_images)
    ids, data_dict = [], []
    def unpack(bids, dets):
        for id_, (bboxes, scores, cls_inds) in zip(bids, dets):
            ids.append(id_)
            for k, box in enumerate(bboxes):
                x1, y1, x2, y2 = float(box[0]), float(box[1]), float(box[2]), float(box[3])
                data_dict.append({"image_id": id_, "category_id": dataset.class_ids[int(cls_inds[k])],
                                  "bbox": [x1, y1, x2 - x1, y2 - y1], "score": float(scores[k])})
    pending = None
    for i0, i1 in _batches(n, batch_size):
        xs, sizes, bids = [], [], []
        for i in range(i0, i1):
            img, id_ = dataset.pull_image(i)
            xs.append(torch.from_numpy(np.ascontiguousarray(transform(img)[0][:, :, (2, 1, 0)])).permute(2, 0, 1))
            sizes.append((img.shape[1], img.shape[0]))
            bids.append(int(id_))
        deliver = _submit(net, torch.stack(xs).float(), np.asarray(sizes, np.float32), **kw)
        if pending is not None:
            unpack(pending[0], pending[1]())
        pending = (bids, deliver)
    if pending is not None:
        unpack(pending[0], pending[1]())
    return ids, data_dict
