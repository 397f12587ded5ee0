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

module.exports = { renderSplit };
