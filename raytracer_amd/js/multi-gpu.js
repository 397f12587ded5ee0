'use strict';
/*
 * Row-split rendering over several GPUs from Node (no counterpart in the reference, which drives
 * one GPU).  The image rows are dealt to the GPUs in 8-row stripes, round-robin; every GPU renders
 * the same frame indices on its own rows with the scene replicated, and the accumulator rows are
 * gathered once at the end.  A pixel depends only on (x, y, width, frameIndex), so the assembled
 * image is bit-identical to what one GPU renders.
 *
 *   renderSplit(scenePath, { width, height, spp, frames, devices: [0, 1, ...], maxBounces })
 *     -> Promise<{ accum: Float32Array (width*height*3, interleaved RGB), rays, ms }>
 *
 * One renderer per entry of `devices` (the same device may be listed more than once), all in this
 * process: the native calls are asynchronous, so the GPUs work concurrently while the single JS
 * thread only submits.  The renderer of devices[0] assembles the image.
 */
const { HipRenderer } = require('./hip-renderer');

async function renderSplit(scenePath, opt) {
    const devices = opt.devices && opt.devices.length ? opt.devices : [0];
    const n = devices.length;
    const canvas = () => ({ width: opt.width, height: opt.height });
    const renderers = [];
    for (let i = 0; i < n; i++) {
        const r = await HipRenderer.Create(canvas(), scenePath, {
            device: devices[i], stripeRows: 8, stripeIndex: i, stripeCount: n, maxBounces: opt.maxBounces,
        });
        if (!r) {
            renderers.forEach((x) => x.destroy());
            throw new Error('could not create the renderer of stripe ' + i + ' on device ' + devices[i]);
        }
        r.setSamplesPerPixel(opt.spp || 1);
        renderers.push(r);
    }
    const t0 = Date.now();
    renderers.forEach((r) => r.renderFrames(opt.frames || 1)); // asynchronous: all GPUs start before any is waited for
    let rays = 0;
    for (let i = 1; i < n; i++) renderers[0].importRows(renderers[i].readOwnedRows(), i);
    const accum = renderers[0].readAccum();
    const ms = Date.now() - t0;
    renderers.forEach((r) => { rays += r.getCounters()[0]; r.destroy(); });
    return { accum, rays, ms };
}

/*
 * Frame-split rendering: frames 1..frames are dealt to the GPUs round-robin, every GPU renders whole
 * images and keeps the plain sum of its frames' colours; the sums are added on the host in renderer
 * order, written into the first renderer and closed there with saturate(sum / (frames + 1)).  Meant for
 * scenes so heavy that a frame should not be cut; approximates the sequential recurrence (not
 * bit-identical: addition order, and no per-frame clamp of pixels whose running mean passes 1).
 *
 *   renderFrameSplit(scenePath, { width, height, spp, frames, devices, maxBounces })
 *     -> Promise<{ accum, rays, ms }>
 */
async function renderFrameSplit(scenePath, opt) {
    const devices = opt.devices && opt.devices.length ? opt.devices : [0];
    const n = devices.length, frames = opt.frames || 1;
    const renderers = [];
    for (let i = 0; i < n; i++) {
        const r = await HipRenderer.Create({ width: opt.width, height: opt.height }, scenePath,
                                           { device: devices[i], maxBounces: opt.maxBounces });
        if (!r) {
            renderers.forEach((x) => x.destroy());
            throw new Error('could not create the renderer on device ' + devices[i]);
        }
        r.setSamplesPerPixel(opt.spp || 1);
        renderers.push(r);
    }
    const t0 = Date.now();
    renderers.forEach((r, i) => {
        const count = Math.floor((frames - 1 - i) / n) + 1; // frames 1 + i, 1 + i + n, ... <= frames
        r.clearAccum();
        if (i < frames) r.renderFrameSums(1 + i, count, n);
    });
    const total = renderers[0].readAccum();
    for (let i = 1; i < n; i++) {
        const part = renderers[i].readAccum();
        for (let k = 0; k < total.length; k++) total[k] = Math.fround(total[k] + part[k]);
    }
    renderers[0].writeAccum(total);
    renderers[0].finishFrameSums(frames);
    const accum = renderers[0].readAccum();
    const ms = Date.now() - t0;
    let rays = 0;
    renderers.forEach((r) => { rays += r.getCounters()[0]; r.destroy(); });
    return { accum, rays, ms };
}

module.exports = { renderSplit, renderFrameSplit };
