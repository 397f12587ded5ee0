#!/usr/bin/env node
'use strict';
/*
 * Headless stand-in for the reference's browser shell (src/main.ts): renders a
 * scene with HipRenderer and prints one JSON line; optionally writes the
 * presented image (PPM or PNG, by file name) and raw read-backs for parity checks.
 *
 *   node cli.js --scene scene.json --width 640 --height 360 --spp 16 --frames 4
 *               [--bounces 6] [--image out.png | --ppm out.ppm] [--preview-every k]
 *               [--apply-model-transforms] [--sphere-lights] [--debug-view] [--dump prefix] [--per-frame]
 *               [--devices 0,1,...] [--split rows|frames]
 *
 * --debug-view draws every triangle's normal marker (the shader's unused debug helper).
 * --devices renders with one renderer per listed GPU, image rows dealt to them in 8-row stripes
 * (multi-gpu.js), and writes the assembled image / accumulator; --split frames deals the frames to the
 * GPUs instead (whole images per GPU, sums added at the end: approximate, see include/rt_hip.h).
 *
 * --preview-every k rewrites the image file after every k frames (progressive preview).
 * --present-every k renders frame by frame, as the reference's render loop does, and fills the canvas
 * image after every k-th frame without waiting for the GPU (k = 1: the reference's canvas pass).
 */
const fs = require('fs');
const { HipRenderer } = require('./hip-renderer');
const { writeImage } = require('./image-writer');

function parseArgs(argv) {
    const a = { width: 1536, height: 864, spp: 1, frames: 1, bounces: 6 }; // canvas 1536x864: index.html:33
    for (let i = 2; i < argv.length; i++) {
        const k = argv[i];
        if (k === '--per-frame') { a.perFrame = true; continue; }
        if (k === '--apply-model-transforms') { a.applyModelTransforms = true; continue; }
        if (k === '--sphere-lights') { a.sphereLights = true; continue; }
        if (k === '--debug-view') { a.debugView = true; continue; }
        const v = argv[++i];
        if (k === '--scene') a.scene = v;
        else if (k === '--width') a.width = parseInt(v, 10);
        else if (k === '--height') a.height = parseInt(v, 10);
        else if (k === '--spp') a.spp = parseInt(v, 10);
        else if (k === '--frames') a.frames = parseInt(v, 10);
        else if (k === '--bounces') a.bounces = parseInt(v, 10);
        else if (k === '--ppm' || k === '--image') a.ppm = v;
        else if (k === '--preview-every') a.previewEvery = parseInt(v, 10);
        else if (k === '--present-every') a.presentEvery = parseInt(v, 10);
        else if (k === '--dump') a.dump = v;
        else if (k === '--split') a.split = v;
        else if (k === '--devices') a.devices = v.split(',').map((x) => parseInt(x, 10));
        else throw new Error('unknown argument ' + k);
    }
    if (!a.scene) throw new Error('--scene is required');
    return a;
}

// several GPUs: rows dealt to one renderer per device, assembled on the first
async function mainSplit(a) {
    const { renderSplit, renderFrameSplit } = require('./multi-gpu');
    const r = await (a.split === 'frames' ? renderFrameSplit : renderSplit)(a.scene, { width: a.width, height: a.height, spp: a.spp, frames: a.frames, devices: a.devices,
                                            maxBounces: a.bounces });
    if (a.ppm) {
        const pixels = new Uint8Array(a.width * a.height * 4);
        for (let i = 0; i < a.width * a.height; i++) {
            for (let c = 0; c < 3; c++) pixels[4 * i + c] = Math.floor(r.accum[3 * i + c] * 255 + 0.5); // 8-bit unorm, no gamma
            pixels[4 * i + 3] = 255;
        }
        writeImage(a.ppm, pixels, a.width, a.height);
    }
    if (a.dump) fs.writeFileSync(a.dump + '.accum.f32', Buffer.from(r.accum.buffer));
    console.log(JSON.stringify({ scene: a.scene, width: a.width, height: a.height, frames: a.frames, spp_per_frame: a.spp,
                                 devices: a.devices, rays: r.rays, seconds: r.ms / 1e3, mrays_per_s: r.rays / r.ms / 1e3,
                                 ms_per_frame: r.ms / a.frames }));
}

async function main() {
    const a = parseArgs(process.argv);
    if (a.devices && a.devices.length > 1) return mainSplit(a);
    const canvas = { width: a.width, height: a.height, pixels: a.ppm || a.presentEvery ? new Uint8Array(a.width * a.height * 4) : null };
    // without --present-every the canvas image is only needed when a file is written: filled on demand, not on every frame
    if (a.presentEvery) a.perFrame = true;
    const renderer = await HipRenderer.Create(canvas, a.scene, { maxBounces: a.bounces, presentEvery: a.presentEvery || (1 << 30),
                                                                 applyModelTransforms: a.applyModelTransforms,
                                                                 sphereLights: a.sphereLights, debugView: a.debugView });
    if (!renderer) process.exit(1);
    renderer.setSamplesPerPixel(a.spp);
    if (a.debugView) renderer.setDebug(true);
    const t0 = process.hrtime.bigint();
    const present = () => {
        renderer.presentNow();
        writeImage(a.ppm, canvas.pixels, a.width, a.height);
    };
    const chunk = a.ppm && a.previewEvery > 0 ? a.previewEvery : a.frames;
    for (let done = 0; done < a.frames;) {
        const n = Math.min(chunk, a.frames - done);
        if (a.perFrame) for (let f = 0; f < n; f++) renderer.render();
        else renderer.renderFrames(n);
        done += n;
        if (a.ppm && done < a.frames) present();
    }
    renderer.sync();
    const seconds = Number(process.hrtime.bigint() - t0) / 1e9;
    const counters = renderer.getCounters();
    const out = {
        scene: a.scene, width: a.width, height: a.height, frames: a.frames, spp_per_frame: renderer.samplesPerPixel,
        frame_index: renderer.frameIndex, rays: counters[0], seconds: seconds,
        mrays_per_s: counters[0] / seconds / 1e6, ms_per_frame: (seconds * 1e3) / a.frames,
        triangle_tests_u32: await renderer.getTriangleTestsSinceLastCheck(),
    };
    if (a.ppm) present();
    if (a.dump) {
        fs.writeFileSync(a.dump + '.accum.f32', Buffer.from(renderer.readAccum().buffer));
        fs.writeFileSync(a.dump + '.ids.i32', Buffer.from(renderer.readIds().buffer));
        fs.writeFileSync(a.dump + '.frame.f16', Buffer.from(renderer.readFrameF16().buffer));
    }
    renderer.destroy();
    console.log(JSON.stringify(out));
}

main().catch((e) => { console.error(e); process.exit(1); });
