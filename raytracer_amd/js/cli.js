#!/usr/bin/env node
'use strict';
/*
 * Headless stand-in for the reference's browser shell (src/main.ts): renders a
 * scene with HipRenderer and prints one JSON line; optionally writes the
 * presented image (PPM or PNG, by file name) and raw read-backs for parity checks.
 *
 *   node cli.js --scene scene.json --width 640 --height 360 --spp 16 --frames 4
 *               [--bounces 6] [--image out.png | --ppm out.ppm] [--preview-every k]
 *               [--apply-model-transforms] [--sphere-lights] [--dump prefix] [--per-frame]
 *
 * --preview-every k rewrites the image file after every k frames (progressive preview).
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
        const v = argv[++i];
        if (k === '--scene') a.scene = v;
        else if (k === '--width') a.width = parseInt(v, 10);
        else if (k === '--height') a.height = parseInt(v, 10);
        else if (k === '--spp') a.spp = parseInt(v, 10);
        else if (k === '--frames') a.frames = parseInt(v, 10);
        else if (k === '--bounces') a.bounces = parseInt(v, 10);
        else if (k === '--ppm' || k === '--image') a.ppm = v;
        else if (k === '--preview-every') a.previewEvery = parseInt(v, 10);
        else if (k === '--dump') a.dump = v;
        else throw new Error('unknown argument ' + k);
    }
    if (!a.scene) throw new Error('--scene is required');
    return a;
}

async function main() {
    const a = parseArgs(process.argv);
    const canvas = { width: a.width, height: a.height, pixels: a.ppm ? new Uint8Array(a.width * a.height * 4) : null };
    // the canvas image is only needed when a file is written: fill it on demand, not on every frame
    const renderer = await HipRenderer.Create(canvas, a.scene, { maxBounces: a.bounces, presentEvery: 1 << 30,
                                                                 applyModelTransforms: a.applyModelTransforms,
                                                                 sphereLights: a.sphereLights });
    if (!renderer) process.exit(1);
    renderer.setSamplesPerPixel(a.spp);
    const t0 = process.hrtime.bigint();
    const present = () => {
        renderer.accumulator.render(null, canvas.pixels);
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
